/*
 * dvbt_oracle.h -- CPU restatement ("oracle") of the gr-dvbt RX hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (dvbt_amd/, the C-ABI
 * library, bench.py's GPU leg) may call into this library; it exists so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check and
 * time the reference algorithm on the CPU.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference).  Pinning status (see DESIGN.md "Oracle pinning"):
 *   - Viterbi ACS/traceback: pinned bit-for-bit against the reference's own
 *     lib/d_viterbi.c + lib/d_tab.c compiled unmodified into oracle/_ref/.
 *   - All other stages: the reference holds no tests, fixtures or golden vectors
 *     (every qa_*.cc / qa_*.py is an empty stub) and their sources need GNU Radio
 *     headers that are absent here, so they cannot be compiled: PARITY UNPINNED
 *     by reference execution; pinned only by the known-answer values captured
 *     from the compiled reference in SURVEY.md Appendix F (tests/test_oracle_kat.py)
 *     and by TX->RX loopback closure.
 */
#ifndef DVBT_ORACLE_H
#define DVBT_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include <complex.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef float complex ocf;   /* same arithmetic as std::complex<float> under gcc */

/* enums == include/dvbt/dvbt_config.h:34-75 (values double as TPS encodings) */
enum { O_QPSK = 0, O_QAM16 = 1, O_QAM64 = 2 };
enum { O_NH = 0, O_ALPHA1, O_ALPHA2, O_ALPHA4 };
enum { O_C1_2 = 0, O_C2_3, O_C3_4, O_C5_6, O_C7_8 };
enum { O_T2k = 0, O_T8k = 1 };
enum { O_G1_32 = 0, O_G1_16, O_G1_8, O_G1_4 };

/* lib/dvbt_config.cc:94-250 */
typedef struct {
  int constellation, hierarchy, code_rate, guard, mode, include_cell_id, cell_id;
  int N, cp, Kmin, Kmax, payload, zeros_left, zeros_right;
  int csize, step, m, alpha, k, n;
  float norm;
  int n_cpilot, n_tps, n_spilot;          /* lib/reference_signals_impl.h:33-39 */
  const int *cpilot, *tps;
} o_cfg;

void o_cfg_init(o_cfg *c, int constellation, int hierarchy, int code_rate, int guard,
                int mode, int include_cell_id, int cell_id);

/* ---- pilots / TPS tables (lib/reference_signals_impl.cc:54-126,334-345) ---- */
void o_prbs_wk(const o_cfg *c, char *wk /* Kmax+1 */);
void o_tps_format(const o_cfg *c, int frame_index, const char *wk, unsigned char *tps68);
int  o_bch_check(const unsigned char *tps68);

/* ---- GF(256)/RS (lib/reed_solomon.cc) ---- */
typedef struct { unsigned char exp[256], log[256], l[256], g[17]; } o_rs;
void o_rs_init(o_rs *rs);
void o_rs_encode(const o_rs *rs, const unsigned char *data239, unsigned char *parity16);
/* compat=0: textbook-correct; compat=1: "lowest-location correction lands on data[0]"
 * as observed for the as-compiled reference. returns rs_decode's value */
int  o_rs_decode(const o_rs *rs, unsigned char *data255, int compat);
/* block body, lib/reed_solomon_dec_impl.cc:77-116: nwords*204 -> nwords*188 */
void o_rs_dec_block(const o_rs *rs, const unsigned char *in, unsigned char *out,
                    size_t nwords, int compat, int *nfail, int *ncorr);

/* ---- symbol interleaver (lib/symbol_inner_interleaver_impl.cc) ---- */
void o_sym_H(const o_cfg *c, int *h /* payload */);
void o_sym_interleave(const o_cfg *c, const int *h, const unsigned char *in,
                      unsigned char *out, int symbol_index, int direction);
/* ---- bit (de)interleaver ---- */
void o_bit_interleave(const o_cfg *c, const unsigned char *in, unsigned char *out, size_t n);
void o_bit_deinterleave(const o_cfg *c, const unsigned char *in, unsigned char *out, size_t n);
/* hierarchical modes: the block's two output streams (lib/bit_inner_deinterleaver_impl.cc:148-184); bits the reference reads from behind its matrix are 0 */
void o_bit_deinterleave_hier(const o_cfg *c, const unsigned char *in, unsigned char *outh, unsigned char *outl, size_t n);

/* ---- constellation (lib/dvbt_demap_impl.cc:117-203) ---- */
void o_constellation(const o_cfg *c, float gain, ocf *points /* csize */);
void o_demap(const o_cfg *c, const ocf *points, const ocf *in, unsigned char *out, size_t n);

/* ---- Viterbi (lib/viterbi_decoder_impl.cc + lib/d_viterbi.c) ---- */
typedef struct {
  unsigned char metric[64], path[64];
  unsigned char pp[24][64];
  int store_pos, ntraceback;
} o_vit_core;
void o_vit_core_init(o_vit_core *v, int ntraceback);
void o_vit_butterfly2(o_vit_core *v, const unsigned char *sym4);
unsigned char o_vit_get_output(o_vit_core *v);

int  o_vit_ntraceback(int code_rate);
const unsigned char *o_vit_puncture(int code_rate, int *len);
/* Block-level restatement of viterbi_decoder_impl::general_work for one reset:
 * in = nsym input bytes (m bits each), bsize as in make(); returns output bytes written
 * (= nblocks*bsize*k/8 - ntraceback). */
size_t o_viterbi_decode(const o_cfg *c, int bsize, const unsigned char *in, size_t nsym,
                        unsigned char *out);
/* the same over exactly nsym input bytes (the depunctured bit count must be a multiple of 16) */
size_t o_viterbi_decode_n(const o_cfg *c, const unsigned char *in, size_t nsym, unsigned char *out);
/* ... with the 64 path metrics (minimum subtracted) right behind the snap_at[i]-th get_output call, 64 bytes each (test instrumentation) */
size_t o_viterbi_decode_snap(const o_cfg *c, const unsigned char *in, size_t nsym, unsigned char *out, const long long *snap_at, int nsnap, unsigned char *snaps);
/* ... and a decoder that takes the stream up in the state init[64] right behind get_output call from_call (0: at in[0]); out[] valid from index from_call - 1 on (test instrumentation) */
size_t o_viterbi_decode_from(const o_cfg *c, const unsigned char *in, size_t nsym, unsigned char *out, long long from_call, const unsigned char *init,
                             const long long *snap_at, int nsnap, unsigned char *snaps);

/* ---- Forney byte de-interleaver (lib/convolutional_deinterleaver_impl.cc) ---- */
/* closed form of the 12 FIFOs starting from all-zero state; n bytes -> n bytes */
void o_conv_deinterleave(const unsigned char *in, unsigned char *out, size_t n);
void o_conv_interleave(const unsigned char *in, unsigned char *out, size_t n);

/* ---- energy dispersal / descramble (next-row component) ---- */
void o_energy_prbs(unsigned char *seq1504); /* xor pattern for one 8-packet group, byte0 of each pkt=0 */
void o_energy_dispersal(const unsigned char *ts, unsigned char *out, size_t npackets);
void o_energy_dispersal_from(const unsigned char *ts, unsigned char *out, size_t npackets, size_t packet0);
size_t o_energy_descramble_groups(const unsigned char *in, size_t ngroups, unsigned char *out);
/* restates energy_descramble_impl::general_work given the whole RS output;
 * returns bytes written */
size_t o_energy_descramble(const unsigned char *in, size_t nitems1504, unsigned char *out);

/* ---- front of the flowgraph: stock rational_resampler_ccc(64,70) + multiply_const; o_resample.c.
 * Third-party (gr-filter, gr-fft; absent, unpinned): PARITY UNPINNED, restated from the GNU Radio 3.7 sources. ---- */
int o_resampler_design(int interp, int decim, int *ri, int *rd, float *taps, int cap);
size_t o_resampler_nout(int interp, int decim, size_t nin);
size_t o_resample_scale(int interp, int decim, float scale, const ocf *in, size_t nin, ocf *out, size_t cap);

/* ---- FFT as used between A1 and A3 (gr::fft::fft_vcc forward, shift=True) ---- */
void o_fft_forward_shift(int N, const ocf *in, ocf *out);
void o_ifft_shift(int N, const ocf *in, ocf *out); /* TX: reverse, shift=True, unnormalised */

/* ---- acquisition (lib/ofdm_sym_acquisition_impl.cc) ---- */
typedef struct o_acq o_acq;
o_acq *o_acq_new(const o_cfg *c, float snr_db);
void   o_acq_free(o_acq *a);
/* one general_work call: `in` must expose 2N+cp samples. returns items produced (0/1),
 * *consumed = samples consumed, *sync_start = 1 if a sync_start tag was attached */
int o_acq_work(o_acq *a, const ocf *in, ocf *out, int *consumed, int *sync_start,
               int *cp_start, float *epsilon);

/* the same when `hist` samples of the stream lie in memory in front of in[0] (GNU Radio's circular buffer keeps what was consumed) */
void o_acq_set_avail(o_acq *a, long long avail);   /* samples of the stream in memory from in[0] of the next call on: reads at and beyond give 0 */
int o_acq_work_hist(o_acq *a, const ocf *in, long long hist, ocf *out, int *consumed, int *sync_start,
                    int *cp_start, float *epsilon);

/* ---- demod_reference_signals (pilot_gen::parse_input + block logic) ---- */
typedef struct o_demod o_demod;
o_demod *o_demod_new(const o_cfg *c);
void     o_demod_free(o_demod *d);
/* one general_work call in the 1-item regime. `in` exposes 2 items (cur,next).
 * returns items produced (0/1); tags via out params. */
int o_demod_work(o_demod *d, const ocf *in, ocf *out, int sync_start_tag,
                 int *superframe_start, int *symbol_index, int *info /* [8] diagnostics */);

/* ---- TX generator (test utility; Appendix E) ---- */
/* ts: npackets*188 bytes (byte0=0x47). Produces whole OFDM symbols only.
 * returns number of complex samples written to iq (<= cap). scale multiplies the IFFT output. */
size_t o_tx_symbols_for_packets(const o_cfg *c, size_t npackets);
size_t o_tx_generate(const o_cfg *c, const unsigned char *ts, size_t npackets, float scale,
                     ocf *iq, size_t cap_samples, ocf *freq_taps /* optional nsym*N or NULL */);
size_t o_tx_generate_from(const o_cfg *c, const unsigned char *ts, size_t npackets, size_t packet0, float scale,
                          ocf *iq, size_t cap_samples, ocf *freq_taps);

/* ---- MODEL (not a restatement of the reference, which has no soft path) of the product's soft-decision kernels: o_soft.c */
void o_soft_demap(const o_cfg *c, const ocf *eq, const float *csi, const int *parity, size_t nsym, signed char *out);
void o_soft_plan(long long total_out, int ntb, int *B_out, int *nsteps_out);
long long o_soft_viterbi(const o_cfg *c, const signed char *soft, long long n_soft, long long total_steps, int B, int nsteps, unsigned char *out);

/* ---- the reference's own SSE2 Viterbi kernels (oracle/_ref), timed natively: decoded Mbit/s, -1 when the library is missing */
double o_ref_viterbi_mbps(const char *so_path, size_t nsym, int ntraceback);
/* the block's decode around the reference's own kernels, natively (the counterpart of o_viterbi_decode_n for comparisons on megabytes) */
size_t o_ref_viterbi_decode_n(const char *so_path, const o_cfg *c, const unsigned char *in, size_t nsym, unsigned char *out);

/* ---- whole RX chain, emulating the GNU Radio flowgraph in the 1-item regime ---- */
typedef struct {
  /* capacities in items; any pointer may be NULL to skip that tap */
  ocf *acq_out;  size_t acq_cap;    size_t acq_n;     /* N per symbol */
  ocf *fft_out;  size_t fft_cap;    size_t fft_n;
  ocf *eq_out;   size_t eq_cap;     size_t eq_n;      /* payload per symbol (after superframe start) */
  unsigned char *demap_out, *symdeint_out, *bitdeint_out; size_t sym_cap; size_t sym_n; /* symbols */
  unsigned char *vit_out; size_t vit_cap; size_t vit_n;     /* bytes */
  unsigned char *deint_out; size_t deint_cap; size_t deint_n; /* bytes */
  unsigned char *rs_out; size_t rs_cap; size_t rs_n;         /* bytes */
  unsigned char *ts_out; size_t ts_cap; size_t ts_n;         /* bytes, after energy_descramble */
  int *cp_start; float *epsilon; int *sym_index; size_t meta_cap; /* per acquired symbol */
  int first_out_symbol;   /* index (in acquired-symbol numbering) of superframe_start */
  int n_acquired;
  int rs_fail, rs_corr;
  double t_stage[10];     /* seconds per stage (acq,fft,demod,demap,symd,bitd,vit,deint,rs,descr) */
  long long ts_first_packet;  /* RS word index (of this run) of the first TS packet */
  long long stream_rs_items;  /* byte de-interleaver items in stream coordinates (== rs_n/1504 unless the run continues a cut stream) */
  int *freq_offset;           /* optional, meta_cap entries: d_freq_offset of every demodulator call (reference_signals_impl.cc:715-744) */
  long long *call_pos;        /* optional, meta_cap entries: sample at which the general_work call that delivered acquired symbol i began */
  unsigned char *sync_flag;   /* optional, meta_cap entries: 1 = the item carries the sync_start tag (first item of a lock period) */
  unsigned char *bitdeint_lp_out; /* optional, sym_cap symbols: the bit de-interleaver's second output (hierarchical modes; port 0 = bitdeint_out feeds the decoder) */
  unsigned char *sf_flag;     /* optional, meta_cap entries: 1 = demod_reference_signals put the superframe_start tag on this acquired item (the start of a lock
                                 period's delivery: demod_reference_signals_impl.cc:118-136), 2 = the item was delivered downstream without one */
} o_rx_taps;

int o_rx_run(const o_cfg *c, const ocf *iq, size_t nsamples, float snr_db, int bsize,
             int rs_compat, o_rx_taps *t);
/* checker of the product's cut mode (include/dvbt_hip.h: dvbt_rx_set_cut); sym_off = 0 is o_rx_run */
int o_rx_run_cut(const o_cfg *c, const ocf *iq, size_t nsamples, float snr_db, int bsize,
                 int rs_compat, long long sym_off, o_rx_taps *t);

#ifdef __cplusplus
}
#endif
#endif
