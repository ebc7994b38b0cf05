/*
 * dvbt_hip.h -- C ABI of libdvbt_hip.so: the MI355X (gfx950) DVB-T receive hot path
 * behind the gr::dvbt block interfaces of BogdanDIA/gr-dvbt.
 *
 * Plain C: opaque handles, POD parameter structs that carry exactly the arguments of the
 * reference's X::make(...), caller-owned in/out buffers (host memory unless a function
 * says "_device"), explicit sideband struct instead of GNU Radio stream tags.
 *
 * One triple per reference block (create / forecast / work / destroy) + the segment API
 * that runs the whole chain device-resident.  Each entry cites the reference interface
 * it replaces (paths relative to the gr-dvbt tree).
 *
 * Threading: like a GNU Radio block, a handle must not be used from two threads at once.
 * Errors: functions return >= 0 (items produced / DVBT_OK) or a negative dvbt_status.
 * The reference never reports errors (SURVEY 8b "Error conventions"); RS failures are
 * likewise silent here (bytes pass through), and are only counted in the report.
 */
#ifndef DVBT_HIP_H
#define DVBT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  DVBT_OK = 0,
  DVBT_ERR_INVALID = -1,      /* bad parameter */
  DVBT_ERR_NO_DEVICE = -2,    /* no HIP device / kernel image not loadable: there is NO CPU fallback */
  DVBT_ERR_HIP = -3,          /* a HIP runtime call failed (message via dvbt_last_error) */
  DVBT_ERR_CAPACITY = -4,     /* caller buffer or handle capacity too small */
  DVBT_ERR_STATE = -5         /* call sequence violates the block's contract */
} dvbt_status;

/* enums: include/dvbt/dvbt_config.h:34-75 (values double as TPS field encodings).  Every function takes them as int.  A
 * translation unit that also includes gr-dvbt's dvbt/dvbt_config.h (the GNU Radio block shells, gr_dvbt_amd/host/gr) defines
 * DVBT_HIP_NO_ENUMS first: that header puts the same type names into the global namespace (:160-167). */
#ifndef DVBT_HIP_NO_ENUMS
typedef enum { DVBT_QPSK = 0, DVBT_QAM16 = 1, DVBT_QAM64 = 2 } dvbt_constellation_t;
typedef enum { DVBT_NH = 0, DVBT_ALPHA1, DVBT_ALPHA2, DVBT_ALPHA4 } dvbt_hierarchy_t;
typedef enum { DVBT_C1_2 = 0, DVBT_C2_3, DVBT_C3_4, DVBT_C5_6, DVBT_C7_8 } dvbt_code_rate_t;
typedef enum { DVBT_T2k = 0, DVBT_T8k = 1 } dvbt_transmission_mode_t;
typedef enum { DVBT_G1_32 = 0, DVBT_G1_16, DVBT_G1_8, DVBT_G1_4 } dvbt_guard_interval_t;
#endif
#define DVBT_AUTO (-1)   /* dvbt_rx_stream_params.rx.constellation / hierarchy / code_rate: take it from the stream's TPS word */

/* sideband: replaces the stream tags of SURVEY Appendix D */
typedef enum { DVBT_TAG_SYNC_START = 1, DVBT_TAG_SUPERFRAME_START = 2, DVBT_TAG_SYMBOL_INDEX = 3 } dvbt_tag_key;
typedef struct { int64_t rel_offset; int32_t key; int32_t value; } dvbt_tag;   /* offset in items, relative to the call's first item */
typedef struct {
  const dvbt_tag *in_tags; int n_in_tags;      /* tags visible in the input window */
  dvbt_tag *out_tags; int out_cap; int n_out_tags;  /* tags the block attaches to its output */
  int n_consumed;                              /* what the block would pass to consume_each() */
} dvbt_sideband;

/* Every block has two work entries.  dvbt_<blk>_work takes HOST buffers, like gr::block::general_work: it copies in, computes, copies
 * out and synchronises.  dvbt_<blk>_work_device takes DEVICE pointers (hipMalloc'd; same item layout) and a hipStream_t (NULL: the
 * device's default stream, on which blocks chained without an explicit stream are ordered): it only enqueues, so adjacent HIP blocks of a flowgraph hand items to each other without crossing PCIe.
 * Tags still travel through the host-side dvbt_sideband.  Two blocks must know data-dependent counts before they return and
 * therefore read a few bytes back and synchronise the stream once per call (ofdm_sym_acquisition: items produced / samples consumed;
 * demod_reference_signals: which items passed the superframe hunt, and their symbol indices); energy_descramble reads the 16 sync
 * positions of its window (NSYNC search).  The others return counts that follow from the call's sizes and tags alone.  A handle is
 * driven through one of the two entries for its whole life (the history of the streaming blocks lives on the device either way). */
const char *dvbt_last_error(void);
/* device buffers for hosts that do not link the HIP runtime themselves (a GNU Radio block shell, gr_dvbt_amd/host/rx_flowgraph_example.cpp):
 * hipMalloc / hipFree / hipMemcpy (synchronous) / hipStreamSynchronize (stream NULL: the whole device) */
void *dvbt_device_malloc(size_t bytes);
void  dvbt_device_free(void *p);
int   dvbt_copy_to_device(void *dst_device, const void *src_host, size_t bytes);
int   dvbt_copy_to_host(void *dst_host, const void *src_device, size_t bytes);
int   dvbt_synchronize(void *stream);
/* page-lock a host buffer that will be handed to the host-pointer entries again and again (hipHostRegister / hipHostUnregister): a GNU Radio shell registers
 * its input and output buffers once (they live as long as the flowgraph); dvbt_<blk>_work (and dvbt_rx_stream_push for calls of 2 MB and more) then let the DMA engines read and write
 * them directly instead of staging every item through a pinned buffer of the handle (one host memcpy each way).  Buffers that are not registered work as before. */
int   dvbt_host_register(void *p, size_t bytes);
int   dvbt_host_unregister(void *p);
int dvbt_device_count(void);                   /* HIP devices visible; <=0 means the library cannot run */
const char *dvbt_version(void);

/* ------------------------------------------------------------------ derived constants
 * replaces dvbt_config::dvbt_config (lib/dvbt_config.cc:94-250, include/dvbt/dvbt_config.h:95-139) */
typedef struct {
  int fft_length, cp_length, Kmax, payload_length, zeros_on_left, m, cr_k, cr_n;
  float norm;
  int ntraceback;                              /* lib/viterbi_decoder_impl.cc:95-124 */
  int info_bits_per_symbol;
} dvbt_dims;
int dvbt_get_dims(int constellation, int hierarchy, int code_rate, int guard_interval,
                  int transmission_mode, dvbt_dims *out);

/* ------------------------------------------------------------------ A1 ofdm_sym_acquisition
 * replaces ofdm_sym_acquisition::make(blocks, fft_length, occupied_tones, cp_length, snr)
 * (include/dvbt/ofdm_sym_acquisition.h:49) and general_work/forecast
 * (lib/ofdm_sym_acquisition_impl.cc:474-481,489-568).  in: cfloat stream; out: items of N cfloat.
 * Unlike the reference (<=1 item per call) a call may produce up to noutput_items items;
 * n_consumed = (N+cp) per symbol attempted. */
typedef struct { int blocks, fft_length, occupied_tones, cp_length; float snr; } dvbt_ofdm_sym_acquisition_params;
typedef struct dvbt_ofdm_sym_acquisition dvbt_ofdm_sym_acquisition;
int  dvbt_ofdm_sym_acquisition_create(const dvbt_ofdm_sym_acquisition_params *p, dvbt_ofdm_sym_acquisition **out);
int  dvbt_ofdm_sym_acquisition_forecast(const dvbt_ofdm_sym_acquisition *h, int noutput_items, int *ninput_required);
int  dvbt_ofdm_sym_acquisition_work(dvbt_ofdm_sym_acquisition *h, int noutput_items, int ninput_items,
                                    const void *in, void *out, dvbt_sideband *sb);
int  dvbt_ofdm_sym_acquisition_work_device(dvbt_ofdm_sym_acquisition *h, int noutput_items, int ninput_items, const void *in_device, void *out_device,
                                            dvbt_sideband *sb, void *stream);
void dvbt_ofdm_sym_acquisition_destroy(dvbt_ofdm_sym_acquisition *h);

/* ------------------------------------------------------------------ A2 forward FFT (stock fft_vxx in the flowgraph)
 * replaces gr::fft::fft_vcc(fft_size, forward=True, rectangular, shift=True)
 * (apps/dvbt_rx_demo*.grc block fft_vxx_0). items of N cfloat in and out. */
typedef struct { int fft_size; int forward; int shift; } dvbt_fft_params;
typedef struct dvbt_fft dvbt_fft;
int  dvbt_fft_create(const dvbt_fft_params *p, dvbt_fft **out);
int  dvbt_fft_forecast(const dvbt_fft *h, int noutput_items, int *ninput_required);
int  dvbt_fft_work(dvbt_fft *h, int noutput_items, int ninput_items, const void *in, void *out, dvbt_sideband *sb);
int  dvbt_fft_work_device(dvbt_fft *h, int noutput_items, int ninput_items, const void *in_device, void *out_device,
                           dvbt_sideband *sb, void *stream);
void dvbt_fft_destroy(dvbt_fft *h);

/* ------------------------------------------------------------------ A3 demod_reference_signals
 * replaces demod_reference_signals::make(itemsize, ninput, noutput, constellation, hierarchy,
 * code_rate_HP, code_rate_LP, guard_interval, transmission_mode, include_cell_id, cell_id)
 * (include/dvbt/demod_reference_signals.h:50-54); general_work lib/demod_reference_signals_impl.cc:97-150.
 * in: items of ninput cfloat (needs noutput_items+1 visible); out: items of noutput cfloat.
 * Emits SUPERFRAME_START once and SYMBOL_INDEX per produced item. */
typedef struct { int itemsize, ninput, noutput, constellation, hierarchy, code_rate_hp, code_rate_lp,
                 guard_interval, transmission_mode, include_cell_id, cell_id; } dvbt_demod_reference_signals_params;
typedef struct dvbt_demod_reference_signals dvbt_demod_reference_signals;
int  dvbt_demod_reference_signals_create(const dvbt_demod_reference_signals_params *p, dvbt_demod_reference_signals **out);
int  dvbt_demod_reference_signals_forecast(const dvbt_demod_reference_signals *h, int noutput_items, int *ninput_required);
int  dvbt_demod_reference_signals_work(dvbt_demod_reference_signals *h, int noutput_items, int ninput_items,
                                       const void *in, void *out, dvbt_sideband *sb);
int  dvbt_demod_reference_signals_work_device(dvbt_demod_reference_signals *h, int noutput_items, int ninput_items, const void *in_device, void *out_device,
                                               dvbt_sideband *sb, void *stream);
void dvbt_demod_reference_signals_destroy(dvbt_demod_reference_signals *h);

/* ------------------------------------------------------------------ A4 dvbt_demap
 * replaces dvbt_demap::make(nsize, constellation, hierarchy, transmission, gain)
 * (include/dvbt/dvbt_demap.h:50); general_work lib/dvbt_demap_impl.cc:217-240. cfloat x nsize -> u8 x nsize */
typedef struct { int nsize, constellation, hierarchy, transmission_mode; float gain; } dvbt_demap_params;
typedef struct dvbt_demap dvbt_demap;
int  dvbt_demap_create(const dvbt_demap_params *p, dvbt_demap **out);
int  dvbt_demap_forecast(const dvbt_demap *h, int noutput_items, int *ninput_required);
int  dvbt_demap_work(dvbt_demap *h, int noutput_items, int ninput_items, const void *in, void *out, dvbt_sideband *sb);
int  dvbt_demap_work_device(dvbt_demap *h, int noutput_items, int ninput_items, const void *in_device, void *out_device,
                             dvbt_sideband *sb, void *stream);
void dvbt_demap_destroy(dvbt_demap *h);

/* ------------------------------------------------------------------ A5 symbol_inner_interleaver
 * replaces symbol_inner_interleaver::make(ninput, transmission, direction)
 * (include/dvbt/symbol_inner_interleaver.h:50-51); general_work lib/symbol_inner_interleaver_impl.cc:161-219.
 * direction 0 (RX) needs one SYMBOL_INDEX tag per item; direction 1 (TX) counts internally. */
typedef struct { int nsize, transmission_mode, direction; } dvbt_symbol_inner_interleaver_params;
typedef struct dvbt_symbol_inner_interleaver dvbt_symbol_inner_interleaver;
int  dvbt_symbol_inner_interleaver_create(const dvbt_symbol_inner_interleaver_params *p, dvbt_symbol_inner_interleaver **out);
int  dvbt_symbol_inner_interleaver_forecast(const dvbt_symbol_inner_interleaver *h, int noutput_items, int *ninput_required);
int  dvbt_symbol_inner_interleaver_work(dvbt_symbol_inner_interleaver *h, int noutput_items, int ninput_items,
                                        const void *in, void *out, dvbt_sideband *sb);
int  dvbt_symbol_inner_interleaver_work_device(dvbt_symbol_inner_interleaver *h, int noutput_items, int ninput_items, const void *in_device, void *out_device,
                                                dvbt_sideband *sb, void *stream);
void dvbt_symbol_inner_interleaver_destroy(dvbt_symbol_inner_interleaver *h);

/* ------------------------------------------------------------------ A6 bit_inner_deinterleaver
 * replaces bit_inner_deinterleaver::make(nsize, constellation, hierarchy, transmission)
 * (include/dvbt/bit_inner_deinterleaver.h:50-51); general_work lib/bit_inner_deinterleaver_impl.cc:120-184.
 * Non-hierarchical: one output stream (work / work_device).  Hierarchical (ALPHA1 / 2 / 4; io_signature (1, 2)): two output streams, the high-priority
 * bytes (2 bits each) and the low-priority ones (:148-184): work_hier / work_hier_device; work / work_device then deliver output 0 alone, as a flowgraph
 * that connects only the first port gets it.  Two things of the reference's hierarchical branch are not reproducible and are defined here: the bits it
 * reads from behind its bit matrix (64-QAM, low-priority bytes i >= 84 of a block, row 5: undefined behaviour there) are 0; hierarchical QPSK is rejected
 * (the reference's constructor divides by d_v - 2 = 0). */
typedef struct { int nsize, constellation, hierarchy, transmission_mode; } dvbt_bit_inner_deinterleaver_params;
typedef struct dvbt_bit_inner_deinterleaver dvbt_bit_inner_deinterleaver;
int  dvbt_bit_inner_deinterleaver_create(const dvbt_bit_inner_deinterleaver_params *p, dvbt_bit_inner_deinterleaver **out);
int  dvbt_bit_inner_deinterleaver_forecast(const dvbt_bit_inner_deinterleaver *h, int noutput_items, int *ninput_required);
int  dvbt_bit_inner_deinterleaver_work(dvbt_bit_inner_deinterleaver *h, int noutput_items, int ninput_items,
                                       const void *in, void *out, dvbt_sideband *sb);
int  dvbt_bit_inner_deinterleaver_work_device(dvbt_bit_inner_deinterleaver *h, int noutput_items, int ninput_items, const void *in_device, void *out_device,
                                               dvbt_sideband *sb, void *stream);
int  dvbt_bit_inner_deinterleaver_work_hier(dvbt_bit_inner_deinterleaver *h, int noutput_items, int ninput_items,
                                            const void *in, void *out_hp, void *out_lp, dvbt_sideband *sb);
int  dvbt_bit_inner_deinterleaver_work_hier_device(dvbt_bit_inner_deinterleaver *h, int noutput_items, int ninput_items, const void *in_device,
                                                   void *out_hp_device, void *out_lp_device, dvbt_sideband *sb, void *stream);
void dvbt_bit_inner_deinterleaver_destroy(dvbt_bit_inner_deinterleaver *h);

/* ------------------------------------------------------------------ A7 viterbi_decoder
 * replaces viterbi_decoder::make(constellation, hierarchy, coderate, bsize, S0, SK)
 * (include/dvbt/viterbi_decoder.h:51-52); general_work lib/viterbi_decoder_impl.cc:192-324 and the
 * SSE2 kernels lib/d_viterbi.c:461-576,680-735.  u8 stream (m bits per byte) -> u8 stream.
 * noutput_items must be a multiple of bsize*k/8 (set_output_multiple, :141). A SUPERFRAME_START tag
 * at rel_offset 0 resets the decoder; at rel_offset>0 the call consumes up to the tag and returns 0 (:213-229).
 * Unlike the reference (file-scope static state) any number of instances may coexist. */
/* what the Viterbi stage's proof + repair passes did (dvbt_rx_params.viterbi_verify; dvbt_rx_viterbi_proof, dvbt_viterbi_decoder_proof) */
typedef struct {
  int64_t chunks;             /* chunk decoders of the launch */
  int64_t decoded_again;      /* chunks the warm-up alone did not prove (decoded again by the parallel repair pass) */
  int64_t sequential;         /* chunks the sequential pass decoded (a repaired decoder that had not merged with the unproven one after a whole chunk: never observed) */
  int64_t not_proven;         /* chunks that are not proven when the launch ends: the final check's count (viterbi_verify >= 1), -1 when it did not run */
} dvbt_viterbi_proof;
typedef struct { int constellation, hierarchy, code_rate, bsize, S0, SK; } dvbt_viterbi_decoder_params;
typedef struct dvbt_viterbi_decoder dvbt_viterbi_decoder;
int  dvbt_viterbi_decoder_create(const dvbt_viterbi_decoder_params *p, dvbt_viterbi_decoder **out);
int  dvbt_viterbi_decoder_forecast(const dvbt_viterbi_decoder *h, int noutput_items, int *ninput_required);
int  dvbt_viterbi_decoder_work(dvbt_viterbi_decoder *h, int noutput_items, int ninput_items,
                               const void *in, void *out, dvbt_sideband *sb);
int  dvbt_viterbi_decoder_work_device(dvbt_viterbi_decoder *h, int noutput_items, int ninput_items, const void *in_device, void *out_device,
                                       dvbt_sideband *sb, void *stream);
/* the chunk decoders' warm-up for the single block (see dvbt_rx_params.viterbi_warm_windows): 0 = the default 72 windows, else a multiple of 24 in [48, 576].
 * Speed only: the block proves and repairs every launch and carries the streaming decoder's state from call to call (dvbt_rx_params.viterbi_verify). */
int  dvbt_viterbi_decoder_set_warm_windows(dvbt_viterbi_decoder *h, int windows);
/* counters of the block's proof + repair passes, summed over its calls since the last reset (chunks, decoded_again, sequential; not_proven = -1: the block runs no final check) */
int  dvbt_viterbi_decoder_proof(dvbt_viterbi_decoder *h, dvbt_viterbi_proof *out);
void dvbt_viterbi_decoder_destroy(dvbt_viterbi_decoder *h);

/* ------------------------------------------------------------------ A8 convolutional_deinterleaver
 * replaces convolutional_deinterleaver::make(nsize(blocks), I, M)
 * (include/dvbt/convolutional_deinterleaver.h:49); general_work lib/convolutional_deinterleaver_impl.cc:93-150.
 * u8 stream -> items of I*blocks bytes; noutput_items must be even (set_output_multiple(2)). */
typedef struct { int blocks, I, M; } dvbt_convolutional_deinterleaver_params;
typedef struct dvbt_convolutional_deinterleaver dvbt_convolutional_deinterleaver;
int  dvbt_convolutional_deinterleaver_create(const dvbt_convolutional_deinterleaver_params *p, dvbt_convolutional_deinterleaver **out);
int  dvbt_convolutional_deinterleaver_forecast(const dvbt_convolutional_deinterleaver *h, int noutput_items, int *ninput_required);
int  dvbt_convolutional_deinterleaver_work(dvbt_convolutional_deinterleaver *h, int noutput_items, int ninput_items,
                                           const void *in, void *out, dvbt_sideband *sb);
int  dvbt_convolutional_deinterleaver_work_device(dvbt_convolutional_deinterleaver *h, int noutput_items, int ninput_items, const void *in_device, void *out_device,
                                                   dvbt_sideband *sb, void *stream);
void dvbt_convolutional_deinterleaver_destroy(dvbt_convolutional_deinterleaver *h);

/* ------------------------------------------------------------------ A9 reed_solomon_dec
 * replaces reed_solomon_dec::make(p, m, gfpoly, n, k, t, s, blocks)
 * (include/dvbt/reed_solomon_dec.h:49); general_work lib/reed_solomon_dec_impl.cc:77-116 and
 * reed_solomon::rs_decode lib/reed_solomon.cc:246-489. items of blocks*(n-s) -> blocks*(k-s) bytes.
 * Only the DVB parameter set (2,8,0x11d,255,239,8,51) is accepted.
 * oracle_compat = 1 reproduces the as-compiled reference quirk (SURVEY 8c last row / B-1). */
typedef struct { int p, m, gfpoly, n, k, t, s, blocks; int oracle_compat; } dvbt_reed_solomon_dec_params;
typedef struct dvbt_reed_solomon_dec dvbt_reed_solomon_dec;
int  dvbt_reed_solomon_dec_create(const dvbt_reed_solomon_dec_params *p, dvbt_reed_solomon_dec **out);
int  dvbt_reed_solomon_dec_forecast(const dvbt_reed_solomon_dec *h, int noutput_items, int *ninput_required);
int  dvbt_reed_solomon_dec_work(dvbt_reed_solomon_dec *h, int noutput_items, int ninput_items,
                                const void *in, void *out, dvbt_sideband *sb);
int  dvbt_reed_solomon_dec_work_device(dvbt_reed_solomon_dec *h, int noutput_items, int ninput_items, const void *in_device, void *out_device,
                                        dvbt_sideband *sb, void *stream);
void dvbt_reed_solomon_dec_destroy(dvbt_reed_solomon_dec *h);

/* ------------------------------------------------------------------ next row: energy_descramble
 * replaces energy_descramble::make(nblocks) (lib/energy_descramble_impl.cc:108-174).
 * items of 8*188 bytes -> u8 stream. */
typedef struct { int nblocks; } dvbt_energy_descramble_params;
typedef struct dvbt_energy_descramble dvbt_energy_descramble;
int  dvbt_energy_descramble_create(const dvbt_energy_descramble_params *p, dvbt_energy_descramble **out);
int  dvbt_energy_descramble_forecast(const dvbt_energy_descramble *h, int noutput_items, int *ninput_required);
int  dvbt_energy_descramble_work(dvbt_energy_descramble *h, int noutput_items, int ninput_items,
                                 const void *in, void *out, dvbt_sideband *sb);
int  dvbt_energy_descramble_work_device(dvbt_energy_descramble *h, int noutput_items, int ninput_items, const void *in_device, void *out_device,
                                         dvbt_sideband *sb, void *stream);
void dvbt_energy_descramble_destroy(dvbt_energy_descramble *h);

/* ------------------------------------------------------------------ next row 2: front-of-chain resample + scale
 * replaces the stock blocks in front of the path in apps/dvbt_rx_demo*.grc:
 *   rational_resampler_xxx_0 (gr::filter rational_resampler_ccc, interp 64, decim 70, taps none, fbw none)
 *   blocks_multiply_const_vxx_0 (complex const 0.0022097087 @2k, 0.00055242272 @8k)
 * cfloat stream at the file rate (10 Msps) -> cfloat stream at the OFDM elementary rate (64/7 Msps), scaled.
 * GNU Radio (gr-filter, gr-fft) is a third-party dependency that is absent from the reference tree and not version
 * pinned: the taps are designed as the 3.7 series does for taps=None (interp/decim reduced by their gcd, fractional_bw
 * 0.4, Kaiser beta 7 windowed sinc: 1149 taps, 32 branches of 36 for 64/70); parity unpinned, float-tolerance tap.
 * scale = 1.0f for the resampler alone.  forecast/work follow rational_resampler_base (noutput*decim/interp + history). */
typedef struct { int interpolation, decimation; float scale; } dvbt_resampler_params;
typedef struct dvbt_resampler dvbt_resampler;
int  dvbt_resampler_create(const dvbt_resampler_params *p, dvbt_resampler **out);
int  dvbt_resampler_forecast(const dvbt_resampler *h, int noutput_items, int *ninput_required);
int  dvbt_resampler_work(dvbt_resampler *h, int noutput_items, int ninput_items,
                         const void *in, void *out, dvbt_sideband *sb);
int  dvbt_resampler_work_device(dvbt_resampler *h, int noutput_items, int ninput_items, const void *in_device, void *out_device,
                                dvbt_sideband *sb, void *stream);
int  dvbt_resampler_get_taps(const dvbt_resampler *h, float *taps, int cap, int *interpolation, int *decimation);  /* returns the tap count */
void dvbt_resampler_destroy(dvbt_resampler *h);

/* ------------------------------------------------------------------ segment API: the whole chain, device resident
 * One segment = a contiguous run of baseband samples (complex64 at the OFDM elementary rate,
 * i.e. the input of ofdm_sym_acquisition in apps/dvbt_rx_demo*.grc).  The segment is processed
 * exactly as the flowgraph would process it from a cold start (acquire, hunt the superframe
 * start, decode), every stage in HBM; the only host<->device traffic is the input (unless
 * already on the device) and the decoded bytes + report. */
typedef struct {
  int constellation, hierarchy, code_rate, guard_interval, transmission_mode, include_cell_id, cell_id;
  float snr_db;            /* ofdm_sym_acquisition snr, 30 in every demo flowgraph */
  int viterbi_bsize;       /* 768 in every demo flowgraph */
  int rs_oracle_compat;    /* see dvbt_reed_solomon_dec_params */
  int descramble;          /* 1: also run energy_descramble and return TS */
  size_t max_samples;      /* capacity: device buffers are sized for this many input samples */
  int device;              /* HIP device ordinal */
  int viterbi_chunk_bytes; /* 0 = default; decoded bytes per wavefront-chunk */
  int resample_interp, resample_decim;   /* 0,0: the segment is already at the OFDM elementary rate (the north-star tap).
                                            64,70: the segment is the 10 Msps file format; resample + scale run first on the device */
  float front_scale;       /* multiply_const of the flowgraph (used only with resampling; 0 = 1.0) */
  int soft_decision;       /* 0: the reference's hard-decision demapper and Viterbi decoder (the parity path).  1: soft decisions (gr-dvbt's TODO.txt:25-26, never
                              built there): per coded bit an 8-bit max-log likelihood ratio weighted with the carrier's channel power, de-interleaved as soft
                              values, decoded by a soft-input Viterbi decoder; everything behind the decoder unchanged.  No reference exists for it: identical
                              TS on a clean loopback, 2-3 dB of gain at the waterfall (tests/test_gpu_soft.py); the chain takes about 1.6x the hard chain's time.
                              The DEMAP / SYMDEINT / BITDEINT taps are not filled in this mode. */
  int hier_stream;         /* hierarchical modes (hierarchy = ALPHA1 / 2 / 4): which output of bit_inner_deinterleaver feeds the Viterbi decoder: 0 = port 0, the
                              high-priority stream (what a flowgraph that connects the block's first output gets), 1 = port 1, the low-priority stream.
                              The decoder behind it is the reference's: it unpacks d_m bits of every byte whatever the stream carries
                              (lib/viterbi_decoder_impl.cc:93,236-243), so -- as with gr-dvbt itself -- no transport stream comes out of a hierarchical
                              transmission; what is reproduced is every block's output */
  int launch_graph;        /* 1: the ~30 launches of dvbt_rx_segment_enqueue_device are captured in a HIP graph the first time a (segment pointer, length, stream, cut) is
                              seen and replayed as ONE graph launch afterwards (a receiver that works through a resident ring of segments sees the same few
                              over and over).  The graph holds what the launch sequence holds: every size that depends on the data lives in the device-side
                              state block, the grids are worst case.  Not used while stage timing is on (dvbt_rx_enable_timing) or with a resampler in front. */
  int front_priority;      /* 1: dvbt_rx_segment_enqueue_device launches a segment's front end (acquisition .. inner de-interleavers) on a stream of the handle's own
                              with the device's highest priority and joins the caller's stream in front of the Viterbi decoder.  With several segments in flight the
                              front end of segment k + 1 then gets the workgroup slots that segment k's decoder gives up instead of queueing behind them
                              (DESIGN.md 8).  Same kernels, same order within a segment, same bytes. */
  int viterbi_warm_windows;/* 0: the default (72).  The Viterbi decoder runs as independent chunk decoders, each started `warm-up` windows (8 trellis steps = one decoded byte
                              each) in front of its chunk from all-zero metrics; a chunk's decoder IS the streaming decoder of lib/d_viterbi.c from the window on at which
                              its state equals its predecessor's -- inside the warm-up or not depends on the INPUT (of 83,000 chunk starts none is late at a pre-Viterbi bit
                              error rate of 1 %, rate 7/8, SURVEY 8d's prescribed level; two at 2 %; about one in a few hundred on a collapsed channel, >= 3 %, and on the
                              hierarchical modes' degenerate decoder input, for up to ~125 windows: tools/hier_warmup.py).  Since round 6 this parameter
                              moves SPEED only: the launch proves every chunk and decodes the late ones again (viterbi_verify below), so the bytes are the streaming
                              decoder's at any warm-up.  A multiple of 24 in [48, 1152]: that many windows instead of 72 (fewer chunks to decode again, more windows per
                              chunk: 144 costs +2.4 % of the decoder's time at the headline's chunk size, 288 +7.9 %).  Hard-decision decoder. */
  int viterbi_verify;      /* The Viterbi stage is the reference's ONE streaming decoder (viterbi_decoder_impl.cc:192-324, d_viterbi.c:680-735) by construction: every chunk
                              decoder leaves its state (64 path metrics + tie-break bits: two registers per lane) as it stands at its chunk's first window, its
                              predecessor leaves its own at the same window; equal states make equal decisions from there on (oracle/o_viterbi.c::o_viterbi_decode_snap +
                              tests/test_viterbi_boundary_proof_model.py hold the criterion on the CPU), and chunk 0 starts the stream.  A checker lists the chunks whose
                              two states differ, a repair pass decodes each of them again from its predecessor's state (no warm-up needed), and where a repaired decoder
                              does not arrive in the state the next chunk was checked against, one sequential decoder walks on in stream order.  All on the device, the
                              host reads no verdict; the passes return at once when every chunk is proven (the usual case: three small launches, +2 % of the decoder).
                                0  (default) proof + repair;  1  the same plus a final check whose count dvbt_rx_viterbi_check / dvbt_rx_viterbi_proof report (0 by
                                construction);  2  proof only, nothing decoded again: the count says how many chunks the warm-up alone does not prove (statistics);
                                3  test hook: the sequential pass does all the repairs;  4  test hook: every repaired chunk hands the chunk behind it to the sequential pass;  -1  the plain chunk decoders (no states, no passes: equal to the streaming decoder
                                where the warm-up suffices, as until round 5).
                              Chunk sizes are rounded up to a multiple of 24 (unless -1).  Hard-decision decoder; every entry (segment API, streaming entry, hierarchical
                              modes -- which ran ONE decoder on one wavefront until round 5).  The single viterbi_decoder block does the same and carries the state from call to call. */
} dvbt_rx_params;

typedef struct {
  int32_t status;              /* 0 ok; bit0: initial acquisition failed; bit1: CP tracking lost;
                                  bit2: no superframe start found;
                                  bit4: the stream's TPS disagrees with the configured parameters (tps_mismatch);
                                  bit7: a segment that continues a cut stream (dvbt_rx_set_cut) lost the CP lock: its Viterbi stream was laid out from its own
                                  lock periods, the cut offset does not apply to its counts (stream_symbol_offset is reported as 0): do not stitch it;
                                  bit8: dvbt_rx_segment_run gave up walking the segment's lock periods (more than 1024 periods or 4096 searches): the rest is not decoded */
  int32_t n_symbols;           /* OFDM symbols acquired */
  int32_t first_out_symbol;    /* symbol at which superframe_start fired, -1 if none */
  int32_t n_out_symbols;       /* symbols passed downstream */
  int32_t cp_start0;           /* d_cp_start after initial acquisition */
  int32_t first_call;          /* general_work call (window of N+cp samples) in which the initial acquisition succeeded */
  int64_t n_viterbi_bytes;
  int64_t n_rs_items;          /* items of 8 codewords */
  int64_t n_rs_bytes;          /* RS words decoded x 188 (= n_rs_items*1504 unless the segment continues a cut stream) */
  int64_t n_ts_bytes;          /* after energy_descramble (0 when descramble==0) */
  int32_t rs_fail_words, rs_corrected_symbols;
  int64_t resume_sample;       /* status bit1 (lock lost): sample of the caller's segment (at the OFDM elementary rate) at which the reference
                                  would start re-acquiring: the call that lost the lock consumes half a window (N+cp)/2; else 0.
                                  dvbt_rx_segment_run restarts there by itself while no superframe start has been found yet (the
                                  start-up transient of a segment that begins with more than one window of silence). */
  int64_t segment_offset;      /* sample of the caller's segment (OFDM elementary rate) at which the reported decode started: 0 unless
                                  dvbt_rx_segment_run restarted; cp_start0, first_call and the CP_START tap count from here */
  /* cut streams (dvbt_rx_set_cut) */
  int64_t stream_symbol_offset; /* the cut's offset, echoed */
  int64_t ts_first_packet;     /* index, among this segment's RS words, of the first packet of the TS tap */
  int64_t stream_rs_items;     /* items of 8 RS words a chain over the whole stream has produced up to this segment's end */
  /* transmission parameters as signalled by the stream (TODO.txt:25-28 "TPS auto-detection"): the TPS word of a frame that passed the
   * BCH check (reference_signals_impl.cc:385-425), decoded per the layout of format_tps_data (:883-916).  The values use the enums above
   * (they double as the TPS encodings).  status bit 4 is raised when they disagree with the handle's parameters: the segment is still
   * decoded as configured (the reference does not look at them either), but its bytes are then meaningless. */
  uint64_t tps_bits;           /* bit i = s_i; s1-s16 (sync word), s23-s24 (frame number) and s54-s67 (parity) cleared */
  int32_t tps_valid;           /* 1: a BCH-valid TPS frame was received and the fields below are set */
  int32_t tps_length_indicator, tps_constellation, tps_hierarchy, tps_code_rate_hp, tps_code_rate_lp,
          tps_guard_interval, tps_transmission_mode, tps_cell_id;
  int32_t tps_mismatch;        /* bit0 constellation, bit1 hierarchy, bit2 HP code rate, bit3 guard interval, bit4 transmission mode differ */
  /* lock periods (dvbt_rx_segment_run / _run_device follow the reference through every loss of the CP lock inside the segment) */
  int32_t n_lock_periods;      /* periods that found a superframe start and delivered items; > 1: the front-end fields above (n_symbols,
                                  first_out_symbol, cp_start0, first_call, segment_offset, the debug taps) describe the LAST one, the stream
                                  fields (n_viterbi_bytes .. n_ts_bytes, the VITERBI/DEINT/RS/TS taps) the whole segment */
  int32_t total_symbols;       /* OFDM symbols acquired over all lock periods */
} dvbt_rx_report;

typedef enum {
  DVBT_TAP_ACQ = 0,        /* cfloat[n_symbols][N]     output of A1 */
  DVBT_TAP_FFT = 1,        /* cfloat[n_symbols][N]     output of A2 (like ACQ/DEMAP/SYMDEINT/DEINT a debug tap: dvbt_rx_enable_taps first) */
  DVBT_TAP_EQ = 2,         /* cfloat[n_out_symbols][payload]  output of A3 (the float-tolerance tap; written only after dvbt_rx_enable_taps) */
  DVBT_TAP_DEMAP = 3,      /* u8[n_out_symbols][payload] */
  DVBT_TAP_SYMDEINT = 4,
  DVBT_TAP_BITDEINT = 5,
  DVBT_TAP_VITERBI = 6,    /* u8[n_viterbi_bytes] */
  DVBT_TAP_DEINT = 7,      /* u8[n_rs_bytes/188*204] */
  DVBT_TAP_RS = 8,         /* u8[n_rs_bytes] */
  DVBT_TAP_TS = 9,         /* u8[n_ts_bytes] */
  DVBT_TAP_CP_START = 10,  /* i32[n_symbols] */
  DVBT_TAP_SYMBOL_INDEX = 11, /* i32[n_symbols - 1] */
  DVBT_TAP_FREQ_OFFSET = 12, /* i32[n_symbols - 1]  integer carrier offset found for each demodulated symbol (reference_signals_impl.cc:715-744) */
  DVBT_TAP_BITDEINT_LP = 13, /* u8[n_out_symbols][payload]  hierarchical modes: the bit de-interleaver's second output (BITDEINT holds the first) */
  DVBT_TAP_SOFT = 14,      /* i8[n_out_symbols][payload * m]  soft-decision mode: the log-likelihood ratio of every coded bit in the decoder's input order (behind both
                              inner de-interleavers); > 0: the bit is more likely 0.  The EQ tap is its input (always filled in this mode) ... */
  DVBT_TAP_BITDEINT_LOG = 16, /* u8[...]  dvbt_rx_enable_taps(h, 2): the Viterbi decoder's input of every lock period (dvbt_rx_period_taps says where) */
  DVBT_TAP_CSI = 15        /* f32[n_out_symbols][payload]  ... together with the channel state of every carrier, 1 / |interpolated equaliser gain|^2 */
} dvbt_tap;

typedef struct dvbt_rx dvbt_rx;
int  dvbt_rx_create(const dvbt_rx_params *p, dvbt_rx **out);
/* Cut streams (the path shards over independent baseband segments).  A long stream is cut near superframe
 * boundaries; every piece is decoded as a segment of its own (other handle, stream or GPU).  A piece that does not hold the
 * beginning of the stream begins a pre-roll before the superframe start it is to deliver (long enough for CP lock and one
 * whole TPS frame) and declares, before it is enqueued, how many OFDM symbols lie between the stream's FIRST superframe
 * start (where the reference's chain starts decoding: demod_reference_signals_impl.cc:118-136) and its own: a multiple of
 * 272.  The library then takes the Viterbi block count (viterbi_decoder_impl.cc:198), the ntraceback delay and the even
 * item count of the byte de-interleaver (convolutional_deinterleaver_impl.cc:55-61) in stream coordinates, and the TS tap
 * starts at the piece's first NSYNC and runs in whole 8-packet groups to the end of its RS words (no two-item hold-back,
 * energy_descramble_impl.cc:139-141: that belongs to the stream's end).  The first RS words of a piece mix the
 * de-interleaver's zero fill with data, as at a stream start: the piece before delivers those packets (post-roll).
 * gr_dvbt_amd/multi.py (plan_cuts / stitch_ts) holds the host side: concatenating the trimmed pieces gives, byte for byte,
 * the TS of one chain over the whole stream.  offset 0 (the default) = the piece holds the beginning of the stream. */
typedef struct {
  int64_t stream_symbol_offset;
  /* the two fields below are what the streaming entry (dvbt_rx_stream_*) knows about the stream a piece continues; 0 / 0 = the plain cut above.
   * start_delay_symbols (0..271): the piece delivers from this many OFDM symbols BEHIND the superframe start its own pilot engine finds.  The reference's
   * demodulator re-hunts the superframe start after a lost CP lock on counters that went on counting through the gap (lib/demod_reference_signals_impl.cc:115-136,
   * the pilot engine's members live on): when no whole TPS frame lies between the new lock and the counters' next "frame d_fi_start, symbol 0", it declares the
   * start there -- a whole number of symbols off the transmitted grid -- and decodes from that symbol until the next loss.  Pieces that continue such a lock
   * period must start where the reference's chain counts from.
   * descr_call_phase: 1 + (first packet of the whole-stream descrambler's two-item calls, mod 16, counted in this piece's RS words); 0 = unknown.  When known
   * the piece checks that every such call inside it finds its NSYNC (lib/energy_descramble_impl.cc:121-141 re-searches otherwise) and reports the outcome
   * to the streaming entry, which then follows the descrambler call by call. */
  int32_t start_delay_symbols;
  int32_t descr_call_phase;
} dvbt_rx_cut;
int  dvbt_rx_set_cut(dvbt_rx *h, const dvbt_rx_cut *cut);   /* applies to the segments enqueued afterwards */
/* iq: host pointer to nsamples complex64 (copied to the device first) */
int  dvbt_rx_segment_run(dvbt_rx *h, const void *iq_host, size_t nsamples, dvbt_rx_report *report);
/* the same for a segment that is already in device memory (stream: a hipStream_t or NULL).  Like dvbt_rx_segment_run it is synchronous
 * and follows the reference through every loss of the CP lock inside the segment (ofdm_sym_acquisition_impl.cc:545-559: half a window
 * consumed, full search again; demod_reference_signals hunts the superframe start again, :115-136; the Viterbi decoder is reset at the
 * new start and the byte de-interleaver realigned, viterbi_decoder_impl.cc:213-229, convolutional_deinterleaver_impl.cc:109-120; the
 * descrambler re-searches its NSYNC, energy_descramble_impl.cc:121-141).  dvbt_rx_segment_enqueue_device below never waits for the
 * device: it decodes the segment's FIRST lock period and reports where the lock ended (status bit 1, resume_sample). */
int  dvbt_rx_segment_run_device(dvbt_rx *h, const void *iq_device, size_t nsamples, void *stream, dvbt_rx_report *report);
/* iq_device: device pointer (hipMalloc'd, e.g. a torch tensor's data_ptr). stream: a hipStream_t or NULL.
 * Asynchronous w.r.t. the host: enqueue only.  Use dvbt_rx_segment_finish to wait and fetch the report. */
int  dvbt_rx_segment_enqueue_device(dvbt_rx *h, const void *iq_device, size_t nsamples, void *stream);
int  dvbt_rx_segment_finish(dvbt_rx *h, dvbt_rx_report *report);
/* the CP-lock periods the last dvbt_rx_segment_run / _run_device walked through (ofdm_sym_acquisition alone, in stream order): where the
 * search that found the lock started, the call and d_cp_start of the initial acquisition, how many symbols the lock held.  A period with
 * n_symbols == 0 is a peak that the tracking search of the same call already missed (ofdm_sym_acquisition_impl.cc:503-559).  Returns the
 * number of periods (may exceed cap). */
typedef struct {
  int64_t offset;             /* sample of the segment (OFDM elementary rate) at which this search began */
  int32_t first_call;         /* window (of N+cp samples, from offset) in which the initial acquisition found its peak */
  int32_t cp_start0;          /* d_cp_start of that call */
  int32_t n_symbols;          /* items the lock delivered before it was lost (or the segment ended) */
  int32_t first_out_symbol;   /* 1 + symbol of the period at which superframe_start fired, 0 if it delivered nothing downstream */
} dvbt_lock_period;
/* what the proof + repair passes of the handle's last launch of the Viterbi decoder did (see dvbt_rx_params.viterbi_verify) */
int  dvbt_rx_viterbi_proof(dvbt_rx *h, dvbt_viterbi_proof *out);
/* the same counters summed over every launch of the handle's decoder since it was created (a synchronous run launches the decoder once per lock period); not_proven = -1 */
int  dvbt_rx_viterbi_proof_total(dvbt_rx *h, dvbt_viterbi_proof *out);
/* (chunks, not_proven) of the same; needs the final check (viterbi_verify >= 1) */
int  dvbt_rx_viterbi_check(dvbt_rx *h, int64_t *chunks, int64_t *unproven);
int  dvbt_rx_lock_periods(dvbt_rx *h, dvbt_lock_period *out, int cap);
/* how the handle's lock-period walks ran so far: acquisition-only passes through the one-launch tracker (acq_small_kernel: look-ahead windows of up to
 * small_max_calls calls, the tracking metric in chunks of small_chunk_calls calls -- 16 for a short guard interval down to 2 for cp = 2048, what fits the LDS) and
 * through the general kernels (long windows, periods the one-launch tracker handed back) */
typedef struct { int64_t small_passes, general_passes; int32_t small_chunk_calls, small_max_calls; } dvbt_walk_stats;
int  dvbt_rx_walk_stats(dvbt_rx *h, dvbt_walk_stats *out);
/* copy a tap of the last finished segment to host memory; returns bytes written */
int64_t dvbt_rx_read_tap(dvbt_rx *h, int tap, void *dst_host, size_t cap_bytes);
/* device pointer of a tap's buffer (for RCCL gathers of the decoded packets) */
void *dvbt_rx_tap_device_ptr(dvbt_rx *h, int tap);
/* average device time in ms of the named stage over the segments finished since the last
 * dvbt_rx_enable_timing(h, 1) (HIP events on the segment's stream); stage names: "acq","fft","demod","inner","viterbi","rs","total".
 * dvbt_rx_enable_timing(h, 2): events around the decoder only ("viterbi"; the other names return -1) -- an event record holds an otherwise
 * idle stream for ~6 us, seven of them are 1 % of a step that runs alone.  0 switches the events off. */
double dvbt_rx_stage_ms(dvbt_rx *h, const char *stage);
int  dvbt_rx_enable_timing(dvbt_rx *h, int enable);
/* allocate (1) / free (0) the debug-only taps ACQ, DEMAP, SYMDEINT, DEINT; the other taps are
 * pipeline buffers and always readable.  Call before the segment whose taps are wanted.
 * 2: the same plus the per-lock-period log of the Viterbi decoder's input (dvbt_rx_period_taps): the BITDEINT tap holds the LAST period of a
 * segment that lost its lock, the log keeps every period's (synchronous entries dvbt_rx_segment_run / _run_device). */
int  dvbt_rx_enable_taps(dvbt_rx *h, int enable);
/* one entry per lock period of the last synchronous run that reached the Viterbi decoder, in stream order: where its decoder input lies in the
 * log (DVBT_TAP_BITDEINT_LOG) and where its decoded bytes lie in the VITERBI tap */
typedef struct { int64_t bitdeint_offset, bitdeint_bytes, viterbi_offset, viterbi_bytes; } dvbt_period_tap;
int  dvbt_rx_period_taps(dvbt_rx *h, dvbt_period_tap *out, int cap);   /* returns the number of entries */
void dvbt_rx_destroy(dvbt_rx *h);

/* ------------------------------------------------------------------ streaming entry: the whole chain behind push / pull
 * What a C++ host (one gr::block whose work() takes cfloat and gives TS bytes, gr_dvbt_amd/host/gr/rx_hip_impl.cc) needs to decode a
 * stream of any length at the segment API's speed: samples are pushed in calls of ANY size (complex64 at the OFDM elementary rate, the
 * input of ofdm_sym_acquisition in apps/dvbt_rx_demo*.grc), the library batches them into pieces of whole superframes with the pre- and
 * post-roll of SURVEY 8e (one CP lock + one whole TPS frame in front of a piece's first superframe boundary, the byte de-interleaver's
 * fill and the distance to the next NSYNC behind its last one), decodes every piece device resident on one of two internal chains (the
 * next piece's samples are copied in while the previous one decodes), trims the pieces and delivers the TS in order.  The bytes are
 * exactly those of ONE chain over the whole stream (dvbt_rx_segment_run on all samples at once; tests/test_gpu_stream.py), i.e. what the
 * reference's flowgraph writes to its file sink: this is gr_dvbt_amd/multi.py's plan_cuts / stitch_plan inside the library.
 * segment_superframes: superframes a piece owns (0 = 16; device memory ~ 2 x (segment_superframes + 3) superframes of samples + the
 * chains' own buffers).  rx.max_samples, rx.resample_* are ignored / must be 0 (no resampler in front).
 * A piece is decoded once the stream is known to go on far enough for the next piece to stand on its own (one superframe + 76 symbols behind its begin), so the
 * TS lags the input by about segment_superframes + 1 superframes.
 * A lost CP lock: the stream follows the reference through it, byte for byte (tests/test_gpu_stream.py, tests/test_gpu_stream_loss.py: dropouts, noise bursts,
 * BASELINE config 5 at 9 dB).  While no lock period is established -- at the stream's beginning and from a piece in which the lock is lost -- the stream is WALKED
 * window by window (the lock-period walk of dvbt_rx_segment_run with the blocks' state carried from window to window: the peak detector's average, the pilot engine's
 * counters that go on counting through the gap, the byte de-interleaver's alignment and contents, the descrambler's call phase), synchronously; once a lock period
 * has held for two superframes behind its superframe start the stream goes back to pieces, cut on the superframe grid the reference's chain counts on -- also when
 * that grid is a whole number of symbols off the transmitted one (a superframe start declared on stale counters, lib/demod_reference_signals_impl.cc:115-136: the
 * reference decodes garbage from there to the next loss, and so does the stream).  status bit 1 reports the loss; dvbt_rx_stream_trace tells what was done.
 * Sharded streams (world > 1): a rank walks a lost lock to the end of its own piece's samples; the result is the single chain's whenever the reference's chain is
 * back in a lock period on the transmitted grid in front of the next piece's first superframe start and its descrambler has found its NSYNC again on the epoch's
 * own 16-packet call grid (status bit 5 otherwise: the other ranks cannot know).  A stream that never establishes a lock period (config 5 at 9 dB) is walked by
 * every rank and delivered by rank 0.
 * At the lock's edge a piece's fresh acquisition may drop -- or not find -- a lock that the chain before it has seen hold; the piece is then started again behind
 * that loss (a few symbols later; the walk takes over beyond six): it is not a loss of the stream and is not reported as one.
 * Memory per stream object (S = segment_superframes, sf = one superframe of samples = 272 (N + cp) x 8 bytes: 18.4 MB at 8k, 4.6 MB at 2k, guard 1/32):
 *   device: per chain (two by default, `chains`) a sample buffer of (S + 3.7) sf + the chain's own buffers (~0.55 x a sample buffer; x 2.2 in soft-decision mode);
 *   S = 16 at 8k: 2 x 362 MB + 2 x ~200 MB.  The first lost lock adds two walk buffers of twice a sample buffer each (2 x 725 MB at S = 16, 8k), and the buffers behind the
 *   walking chain's Viterbi decoder grow to what a window of that size can lay out (three more buffers of ~2.2 x a piece's decoded bytes: +115 MB at S = 16, 8k QAM64 7/8).
 *   A walk's window is bounded by those buffers: a lock that holds for more than two pieces and a threshold of samples (2 x the sample buffer + (STREAM_PRE + 272) symbols)
 *   without ever reaching a superframe start -- no decodable TPS word: nothing the reference would deliver either -- is cut: the walk starts afresh two superframes back, status bits
 *   2 and 5 are raised and dvbt_rx_stream_trace says so.  With device output (dvbt_rx_stream_set_device_output) a piece's packets are handed over in the chain's TS buffer and the chain
 *   takes a spare one: up to one more TS buffer (~0.25 x a sample buffer) per piece whose packets wait for an exchange step;
 *   page-locked host: the TS ring (ts_ring_bytes, default 96 MB; a consumer that falls further behind is served from the heap) and 32 MB of staging for host
 *   pushes below 2 MB; both taken at create (page-locking them takes ~10 ms: set-up, not the first piece's latency).
 * Threading: like every handle, one thread at a time. */
/* TPS auto-configuration (gr-dvbt's TODO.txt:28 "Autodetect transmission params"): rx.constellation, rx.hierarchy and / or rx.code_rate = DVBT_AUTO.  The
 * transmission mode and the guard interval must be given (the front end is built from them); everything else the stream says itself: the head of the stream
 * (two TPS frames and a margin: 160 symbols + one window) is held back, a probe chain reads a BCH-valid TPS word out of it (lib/reference_signals_impl.cc:385-425,
 * field layout :883-916), the chains are built for what it names and the head is replayed into them -- the TS is that of a stream created with those
 * parameters (tests/test_gpu_tps.py).  While no valid word has been seen the look-ahead doubles (up to four attempts); then push fails with DVBT_ERR_STATE. */
typedef struct {
  dvbt_rx_params rx; int segment_superframes;
  /* sharding over the GPUs of a node, one stream object per process / GPU: every rank is pushed the SAME stream; piece k >= 1 belongs to rank
   * k % world, which alone copies its samples to the device and decodes it (piece 0, the plan, is decoded by every rank and delivered by rank 0).  A rank
   * pulls its own packets with their index in the stream (dvbt_rx_stream_pull_chunk); all ranks' chunks ordered by that index are the single chain's TS:
   * gathering them is the design's one exchange step (RCCL over xGMI in bench.py / gr_dvbt_amd/multi.py).  world = 0 or 1: no sharding. */
  int rank, world;
  /* page-locked ring the decoded TS waits in until it is pulled (0 = 96 MB, ~24 s of the fastest DVB-T transport stream); a consumer that falls further behind
   * than this -- and a stream that cannot have the ring -- is served from heap chunks; dvbt_rx_stream_set_device_output gives it back */
  int64_t ts_ring_bytes;
  /* 1: the samples handed to dvbt_rx_stream_push_device are BORROWED, not copied: the stream remembers where they are and reads them there -- a piece that lies in one
   * contiguous stretch of the caller's memory is decoded in place, a piece across two stretches and the windows of a walk are gathered when they are needed.  The caller keeps
   * the samples valid and unchanged until dvbt_rx_stream_info.samples_released has passed them (a ring of segments resident in HBM does).  dvbt_rx_stream_push (host memory)
   * is refused on such a stream; not together with DVBT_AUTO.  0: every push copies (the default: the caller's buffer is free when the call returns). */
  int borrow_device_pushes;
  /* chains (handle + sample buffer + HIP stream) of the stream object: 0 = 2 (the default), or 2 .. 4.  The pieces take them in turn; `chains - 1` pieces decode while the next one
   * fills, and the small latency-bound kernels of the later pieces run in the gaps of the earlier pieces' decoders (the segment API's "several segments in flight"):
   * 4 chains: 4.3 instead of 4.5 ms per 64-superframe piece at the headline workload -- for twice the device memory (every chain has its sample buffer and its buffers behind the
   * decoder; the walk's buffers hold `chains` sample buffers and a threshold).  Same bytes. */
  int chains;
} dvbt_rx_stream_params;
typedef struct {
  int32_t status;              /* dvbt_rx_report.status bits of the pieces, OR-ed (bit 1 only when the lock was lost inside a piece) | bit 5: a piece
                                  delivered fewer packets than its span of the stream | bit 6: a piece's sync byte was not where the stream's packet
                                  count puts it */
  int32_t pieces_in_flight, finished;
  int64_t samples_pushed, ts_bytes_decoded, ts_bytes_ready, ts_bytes_pulled;
  int64_t first_superframe_call;   /* call (window of N+cp samples) of the stream's first superframe start, -1 before it is known */
  int64_t first_ts_packet;         /* RS word (counted from that superframe start) of the first TS packet, -1 before it is known */
  int32_t constellation, hierarchy, code_rate;   /* the parameters the chains run with; -1 while they are being detected (DVBT_AUTO) */
  int32_t auto_configured;         /* 1: they were taken from the stream's TPS word */
  int64_t samples_released;        /* borrow_device_pushes: the stream will not read samples in front of this stream position again (else = samples_pushed) */
  int32_t in_walk;                 /* 1: no lock period is established (the stream's beginning, or behind a lost CP lock): the stream is being walked window by window */
} dvbt_rx_stream_info;
typedef struct dvbt_rx_stream dvbt_rx_stream;
int  dvbt_rx_stream_create(const dvbt_rx_stream_params *p, dvbt_rx_stream **out);
/* iq_host: nsamples complex64 in host memory; they have left the caller's buffer when the call returns (the decode has not finished) */
int  dvbt_rx_stream_push(dvbt_rx_stream *s, const void *iq_host, size_t nsamples);
/* the same for samples in device memory, ordered behind `stream` (a hipStream_t or NULL); the caller keeps them valid until that copy has run */
int  dvbt_rx_stream_push_device(dvbt_rx_stream *s, const void *iq_device, size_t nsamples, void *stream);
/* TS bytes that are ready, in stream order, up to cap; never waits for the device before dvbt_rx_stream_finish.  Returns the bytes written */
int64_t dvbt_rx_stream_pull(dvbt_rx_stream *s, void *ts_host, size_t cap);
/* one contiguous run of this rank's packets (never across two pieces) and the index of its first packet in the stream's TS (packet 0 = the first packet
 * the single chain delivers is *first_packet == first_ts_packet of dvbt_rx_stream_info): what a sharded stream's ranks exchange.  cap >= 188 */
int64_t dvbt_rx_stream_pull_chunk(dvbt_rx_stream *s, void *ts_host, size_t cap, int64_t *first_packet);
/* end of the stream: decodes what is left (energy_descramble's two-item hold-back and the block roundings apply here, as at the end of
 * the reference's run) and waits for it; pull then drains the rest */
int  dvbt_rx_stream_finish(dvbt_rx_stream *s);
int  dvbt_rx_stream_status(const dvbt_rx_stream *s, dvbt_rx_stream_info *info);
/* what the stream did, as text: one line per walk window, per epoch that was established, per piece that left its epoch (a lost CP lock) or whose descrambler
 * had to be followed on the host; returns the trace's length (the text is cut to cap - 1 characters, 0-terminated).  Bounded at 64 KB. */
int64_t dvbt_rx_stream_trace(const dvbt_rx_stream *s, char *dst, size_t cap);
/* the Viterbi stage's proof + repair passes (dvbt_rx_params.viterbi_verify: the stream's chains run them like every handle) summed over the stream's launches so far */
int  dvbt_rx_stream_viterbi_proof(const dvbt_rx_stream *s, dvbt_viterbi_proof *out);
void dvbt_rx_stream_destroy(dvbt_rx_stream *s);

/* ------------------------------------------------------------------ the exchange step of a sharded stream
 * north_star: "C++ host code ... a single RCCL gather of decoded TS packets over xGMI".  One process per GPU; every process creates a dvbt_rx_stream with its
 * rank / world, is pushed the same stream and decodes its own pieces (no data-path collective).  Per step, dvbt_rx_stream_gather moves every rank's next run
 * of finished packets to `root` in ONE group of ncclSend / ncclRecv on device buffers (a gather of fixed-stride slots: 64-byte header {first packet index,
 * byte count, drained flag, error flag} + packets, the layout of gr_dvbt_amd/multi.py; the headers go to every rank in the same group) and hands root the runs
 * with their packet indices; ordered by that index they are the single chain's TS.  librccl.so is opened with dlopen at first use (no link-time dependency).  The communicator can come from here
 * (dvbt_rccl_unique_id on one rank, its 128 bytes carried to the others by the host -- a file, a socket, MPI --, then dvbt_rccl_comm_create everywhere:
 * ncclGetUniqueId / ncclCommInitRank), so a host needs no RCCL headers.  gr_dvbt_amd/host/rx_multi_example.cpp is such a host. */
typedef struct dvbt_rccl_comm dvbt_rccl_comm;
int  dvbt_rccl_unique_id(void *id128_out);                                    /* 128 bytes */
int  dvbt_rccl_comm_create(const void *id128, int rank, int world, int device, dvbt_rccl_comm **out);   /* collective over the `world` processes */
void dvbt_rccl_comm_destroy(dvbt_rccl_comm *c);
typedef struct { int64_t first_packet; int64_t nbytes; int64_t offset; } dvbt_gather_chunk;   /* rank r's run: packet index in the stream's TS, bytes, where they sit in ts_host */
/* The step, asynchronous and double-buffered (what a host that wants the exchange behind its decode calls; bench.py's Python path does the same with
 * torch.distributed).  dvbt_rx_stream_set_device_output (before the first push): the decoded TS stays in device memory (a ring of ring_bytes, 0 = 64 MB; pull /
 * pull_chunk then refuse; a piece's packets wait in the very buffer they were decoded into, ring_bytes holds the walk's smaller runs) and a step copies its run device to
 * device into the send slot -- no PCIe hop on any rank but the root's one download (none with DVBT_GATHER_DEVICE, below).
 * dvbt_rx_stream_gather_enqueue: collective (the same root and slot_packets on every rank; a rank gives at most slot_packets packets per step); issues ONE group of
 * ncclSend / ncclRecv on the communicator's own HIP stream -- the slot to the root, the 64-byte header to every rank -- and returns at once; at most two steps in
 * flight.  dvbt_rx_stream_gather_wait: the oldest step in flight.  root: ts_host (cap >= world * slot_packets * 188) receives the runs in rank order, chunks[world]
 * describes them; returns the bytes written.  Others: ts_host / chunks may be NULL; returns 0.  *all_done (every rank, from the headers of the same step) = 1 once
 * every rank's stream is finished and drained.  A rank that fails in front of a step's group sends an empty slot with an error flag: the step completes
 * everywhere and fails on every rank (DVBT_ERR_STATE) -- no rank is left waiting.  Receive space is allocated on the root only. */
int  dvbt_rx_stream_set_device_output(dvbt_rx_stream *s, size_t ring_bytes);
int  dvbt_rx_stream_gather_enqueue(dvbt_rx_stream *s, dvbt_rccl_comm *c, int root, int slot_packets);
int64_t dvbt_rx_stream_gather_wait(dvbt_rx_stream *s, dvbt_rccl_comm *c, void *ts_host, size_t cap, dvbt_gather_chunk *chunks, int *all_done);
/* root, after a wait with ts_host = NULL: the runs stay where the step's download put them -- chunks[r].offset then counts from this page-locked buffer (valid
 * until the second next wait): a root that writes the packets on needs no copy of its own (at the headline rate the TS is ~15 GB/s).  The download copies, per
 * rank, the header and the bytes the header declares -- not the slot. */
const void *dvbt_rccl_step_buffer(const dvbt_rccl_comm *c);
/* The step that leaves the runs in the ROOT'S DEVICE MEMORY (north_star: the clock ends at "last TS byte resident on rank 0"): dvbt_rx_stream_gather_enqueue_ex with
 * DVBT_GATHER_DEVICE downloads nothing but the 64-byte headers; dvbt_rx_stream_gather_wait (ts_host = NULL) describes the runs in chunks[], offsets counted from
 * dvbt_rccl_step_device_buffer (device pointer, valid until the second next wait).  flags = 0: dvbt_rx_stream_gather_enqueue.  Same flags on every rank. */
#define DVBT_GATHER_DEVICE 1
int  dvbt_rx_stream_gather_enqueue_ex(dvbt_rx_stream *s, dvbt_rccl_comm *c, int root, int slot_packets, int flags);
const void *dvbt_rccl_step_device_buffer(const dvbt_rccl_comm *c);
/* the exchange buffers for steps of slot_packets packets towards root (flags as above: without DVBT_GATHER_DEVICE the root also gets its page-locked mirror).  The
 * enqueue calls it itself; called first, on every rank, it moves an allocation failure in front of the first step (inside a step such a failure becomes an error-flag
 * slot sent from the stream's sample buffer, so that no rank waits -- see dvbt_rccl.inc) */
int  dvbt_rccl_comm_reserve(dvbt_rccl_comm *c, int root, int slot_packets, int flags);
/* the two at once (blocking): one group and one synchronisation per step */
int64_t dvbt_rx_stream_gather(dvbt_rx_stream *s, dvbt_rccl_comm *c, int root, int slot_packets, void *ts_host, size_t cap, dvbt_gather_chunk *chunks, int *all_done);

/* ------------------------------------------------------------------ test hooks (used by tests/ only)
 * the two peak detectors of the acquisition's trackers (lib/ofdm_sym_acquisition_impl.cc:72-146 restated sample by sample, and the wavefront-wide form the
 * sequential trackers run) on n cases of 16 metric values + a carried d_avg each: out[4k] npk, out[4k+1] position (-1: none) of the first, out[4k+2..3]
 * of the second; avg_out[2k], avg_out[2k+1]: d_avg after the window */
int dvbt_debug_peak_detect(const float *lambda_host, const float *avg_host, int n, int32_t *out_host, float *avg_out_host);
/* the streaming entry's host-side follower of energy_descramble (lib/energy_descramble_impl.cc:121-141; csrc/dvbt_stream.inc::descr_walk), which a stream uses wherever the
 * descrambler re-searches its NSYNC: rs = nitems items of 1504 bytes as they leave reed_solomon_dec; the follower is called once per entry of windows[] (items visible so
 * far, ascending: the state it carries from window to window is what is under test) and reports the calls it delivers as runs of (first packet, packets) in runs[2 k],
 * runs[2 k + 1].  Returns the number of runs, or a negative error.  No device needed. */
int64_t dvbt_debug_descr_follow(const uint8_t *rs, size_t nitems, const int64_t *windows, int nwindows, int64_t *runs, size_t cap_runs);
/* the exchange step's world > 1 logic on ONE device (RCCL refuses two ranks on a device, a test box has one): a loopback transport whose ranks are threads of one process -- a send
 * is a posted message, a receive copies device to device behind the sender's stream, a group ends when everything posted has been taken.  *group: NULL on the first call (created and
 * returned), the returned value for the other ranks.  dvbt_rccl_debug_fail_steps: the communicator's next n steps behave as if their exchange buffers could not be allocated (the
 * error-flag slot, sent from the stream's sample buffer).  tests/test_gpu_rccl.py::test_exchange_step_world_2_over_the_loopback_transport */
int dvbt_rccl_comm_create_loopback(void **group, int rank, int world, int device, dvbt_rccl_comm **out);
int dvbt_rccl_debug_fail_steps(dvbt_rccl_comm *c, int n);

#ifdef __cplusplus
}
#endif
#endif /* DVBT_HIP_H */
