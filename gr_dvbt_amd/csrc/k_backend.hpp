// k_backend.hpp -- gfx950 kernels for A4..A9 (+ energy_descramble): integer/byte stages,
// bit-exact against the reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "k_frontend.hpp"

namespace dvbt {

// ---------------------------------------------------------------- A4+A5+A6 fused: demap -> symbol de-interleave -> bit de-interleave
// One workgroup per OFDM symbol, the 6048 (1512) labels staged in LDS.
//   A4 dvbt_demap_impl.cc:167-203  : first strict minimum of (dr*dr + di*di) over the label table
//   A5 symbol_inner_interleaver_impl.cc:202-208 : even symbol out[q]=in[H(q)], odd out[H(q)]=in[q]
//   A6 bit_inner_deinterleaver_impl.cc:138-157 : out[i] bit k = bit (v-1-e) of in[(i-off_e) mod 126], e = perm(k)
struct InnerParams { int payload, m, csize; int nlev; float inv_step, guard; float hshift; int hier; };
// nlev: levels per axis of the constellation's grid (0: no grid, exhaustive search); inv_step = 1/(level spacing); guard: |component| above which the
// exhaustive search is used.  hshift: hierarchical constellations (alpha = 2, 4: dvbt_config.cc:213-225, dvbt_demap_impl.cc:141-142) are the uniform grid
// with the two halves of each axis pushed (alpha - 1) units apart; the grid CELL of a carrier is found on |x| - hshift (clamped at 0), the distances are
// always taken to the true points.  hier: the bit de-interleaver's hierarchical form (two outputs, bit_inner_deinterleaver_impl.cc:148-184)
__device__ __forceinline__ float hier_unshift(float x, float hshift) { return copysignf(fmaxf(fabsf(x) - hshift, 0.f), x); }

// exhaustive search: literally dvbt_demap_impl.cc:167-203
__device__ __forceinline__ int demap_all(float2 v, const float2 *pts, int csize)
{
  float dr = v.x - pts[0].x, di = v.y - pts[0].y;
  float best = dr * dr + di * di; int idx = 0;
  for (int j = 1; j < csize; j++) {
    dr = v.x - pts[j].x; di = v.y - pts[j].y;
    float d = dr * dr + di * di;
    if (d < best) { best = d; idx = j; }
  }
  return idx;
}

// Same result from 4 candidates.  d(i,q) = fl(fl(dx_i^2) + fl(dy_q^2)) is monotone in each addend, so the minimum
// over the table is attained on the per-axis nearest level; the only points that can TIE with it in float are the
// ones whose axis distance is nearly equal, i.e. the second-nearest level on the side of the sample (any other level
// is at least one spacing farther: its square differs by >= 0.75 spacing^2, which rounding cannot absorb below the
// guard magnitude).  So the first strict minimum of the 64-point search (dvbt_demap_impl.cc:167-203) = the smallest
// label among the minima of {nearest, second nearest} x {nearest, second nearest}, all four computed exactly as the
// reference computes them.
// returns -1 when the sample is outside the range where the 4-candidate argument holds (caller falls back to demap_all)
__device__ __forceinline__ int demap_fast(float2 v, const float2 *pts, const unsigned char *label_of, const InnerParams &p)
{
  const bool in_range = p.nlev != 0 && fabsf(v.x) < p.guard && fabsf(v.y) < p.guard;
  const int n = p.nlev;
  // the carrier's cell (hierarchical: on the coordinate with the centre gap closed; a carrier inside the gap lands on the cell boundary at the centre, and the
  // second candidate below -- chosen against the TRUE point -- is then the inner level of the other side: both are evaluated exactly)
  const float gx = p.hshift != 0.f ? hier_unshift(v.x, p.hshift) : v.x, gy = p.hshift != 0.f ? hier_unshift(v.y, p.hshift) : v.y;
  int ji = (int)floorf(gx * p.inv_step + 0.5f * (float)n), jq = (int)floorf(gy * p.inv_step + 0.5f * (float)n);
  ji = ji < 0 ? 0 : ji > n - 1 ? n - 1 : ji; jq = jq < 0 ? 0 : jq > n - 1 ? n - 1 : jq;
  const int L00 = label_of[ji * 8 + jq];
  const float px0 = pts[L00].x, py0 = pts[L00].y;
  int ji2 = ji + (v.x > px0 ? 1 : -1), jq2 = jq + (v.y > py0 ? 1 : -1);
  ji2 = (ji2 < 0 || ji2 > n - 1) ? ji : ji2; jq2 = (jq2 < 0 || jq2 > n - 1) ? jq : jq2;
  const int L10 = label_of[ji2 * 8 + jq], L01 = label_of[ji * 8 + jq2], L11 = label_of[ji2 * 8 + jq2];
  const float px1 = pts[L10].x, py1 = pts[L01].y;
  const float dx0 = v.x - px0, dx1 = v.x - px1, dy0 = v.y - py0, dy1 = v.y - py1;
  const float ax0 = dx0 * dx0, ax1 = dx1 * dx1, ay0 = dy0 * dy0, ay1 = dy1 * dy1;
  const float d00 = ax0 + ay0, d01 = ax0 + ay1, d10 = ax1 + ay0, d11 = ax1 + ay1;
  const float best = fminf(fminf(d00, d01), fminf(d10, d11));
  int idx = 255;
  idx = d00 == best ? min(idx, L00) : idx; idx = d01 == best ? min(idx, L01) : idx;
  idx = d10 == best ? min(idx, L10) : idx; idx = d11 == best ? min(idx, L11) : idx;
  return in_range ? idx : -1;
}
__device__ __forceinline__ int demap_one(float2 v, const float2 *pts, const unsigned char *label_of, const InnerParams &p)
{
  const int f = demap_fast(v, pts, label_of, p);
  return f >= 0 ? f : demap_all(v, pts, p.csize);
}

// LDS image of a symbol's labels: blocks of 126 bytes (the bit interleaver's block) at a stride of 132, each followed
// by a copy of its first 6 bytes, so that 4 consecutive bytes of a cyclically rotated block are one unaligned word
constexpr int IB = 126, IBS = 132;
__host__ __device__ inline size_t inner_lds_bytes(size_t payload) { return ((((payload + IB - 1) / IB) * IBS + 15) & ~(size_t)15) + 64 * 8 + 64; }
__device__ __forceinline__ int ipos(int d) { const int b = d / IB; return d + b * (IBS - IB); }

// mode bits: 1 = demap, 2 = symbol de-interleave, 4 = bit de-interleave (7 = fused chain path).
// Reads are in carrier order (coalesced); the symbol permutation is applied on the LDS store:
// even symbol out[q] = in[H(q)]  <=>  label of source p lands at Hinv[p];  odd: at H[p].
#ifndef DVBT_INNER_THREADS
#define DVBT_INNER_THREADS 256
#endif
// A6 on the padded LDS image: output byte i of a block, bit k (MSB first) = bit (m-1-e) of input byte (i - off_e) mod 126,
// e = perm(k) (bit_inner_deinterleaver_impl.cc:91-99,138-157).  One item = 4 consecutive output bytes: per bit plane one
// (unaligned) word of the rotated block, masked and shifted into place.  VB = bits per carrier (compile time: the plane
// permutation and the rotations are constants).
template <int VB> __device__ __forceinline__ void bit_deint_words(const uint8_t *v, uint8_t *o, int payload, int tid)
{
  constexpr int hv = VB / 2;
  const int nitems = (payload / 126) * 32;                       // 32 words cover the 126 bytes of a block
  for (int it = tid; it < nitems; it += DVBT_INNER_THREADS) {
    const int blk = it >> 5, j = it & 31;
    const unsigned *bw = reinterpret_cast<const unsigned *>(v + blk * 132);
    unsigned val = 0;
#pragma unroll
    for (int k = 0; k < VB; k++) {
      constexpr int offs[6] = {0, 63, 105, 42, 21, 84};
      const int eidx = k / hv + 2 * (k % hv);                     // d_perm, non-hierarchical
      int w = 4 * j - offs[eidx]; if (w < 0) w += 126;
      const unsigned lo = bw[w >> 2], hi = bw[(w >> 2) + 1];
      const unsigned word = __builtin_amdgcn_alignbyte(hi, lo, (unsigned)(w & 3));
      val |= ((word >> (VB - 1 - eidx)) & 0x01010101u) << (VB - 1 - k);
    }
    uint8_t *dst = o + blk * 126 + 4 * j;                         // even address
    *reinterpret_cast<uint16_t *>(dst) = (uint16_t)val;
    if (j < 31) *reinterpret_cast<uint16_t *>(dst + 2) = (uint16_t)(val >> 16);
  }
}

// A6, hierarchical modes (bit_inner_deinterleaver_impl.cc:91-99,148-184): two output streams.  The reference indexes its bit matrix d_b[v][126] with second
// indices beyond 125; the matrix is contiguous, so d_b[e][j] is flat element f = e * 126 + j = bit (v - 1 - f / 126) of input byte (f % 126 - off) mod 126.
// HP byte i: bits f_k = ((v i + k) % 2) * 126 + (v i + k) / 2, k = 0, 1.  LP byte i: k = 2 .. v - 3 (none for 16-QAM: 0; two bits for 64-QAM) with
// f_k = d_perm[v i + k] * 126 + (v i + k) / (v - 2); an f behind the matrix (row 5, i >= 84) is undefined behaviour in the reference (it reads its stack): 0 here,
// as in oracle/o_inner.c::o_bit_deinterleave_hier.  Byte-wise: the hierarchical modes are on no throughput path.
__device__ __forceinline__ void bit_deint_hier(const uint8_t *v, uint8_t *o_hp, uint8_t *o_lp, int payload, int m, int tid)
{
  constexpr int offs[6] = {0, 63, 105, 42, 21, 84};
  for (int idx = tid; idx < payload; idx += DVBT_INNER_THREADS) {
    const int blk = idx / 126, i = idx - blk * 126;
    const uint8_t *bb = v + blk * 132;
    auto bit = [&](int f) -> int { const int e = f / 126, j = f - e * 126; int w = j - offs[e]; if (w < 0) w += 126; return (bb[w] >> (m - 1 - e)) & 1; };
    const int t0 = m * i;
    const int hp = (bit((t0 & 1) * 126 + t0 / 2) << 1) | bit(((t0 + 1) & 1) * 126 + (t0 + 1) / 2);
    int lp = 0;
    for (int k = 2; k < m - 2; k++) {
      const int t = t0 + k, e = (t % (m - 2)) / ((m - 2) / 2) + 2 * (t % ((m - 2) / 2)) + 2, f = e * 126 + t / (m - 2);
      lp = (lp << 1) | (f < m * 126 ? bit(f) : 0);
    }
    o_hp[idx] = (uint8_t)hp;
    if (o_lp) o_lp[idx] = (uint8_t)lp;
  }
}

constexpr int INNER_NB = 12;                         // carriers per thread and batch
constexpr int INNER_THREADS = DVBT_INNER_THREADS;   // workgroup size of inner_kernel: a symbol's 6048 carriers are few serial steps per thread
template <int MODE> __global__ __launch_bounds__(INNER_THREADS) void inner_kernel(const float2 *__restrict__ eq, const uint8_t *__restrict__ in_bytes, InnerParams p,
                                                   const RxState *st, int nitems_fixed, const int *__restrict__ sym_index,
                                                   const float2 *__restrict__ points, const unsigned char *__restrict__ label_tab,
                                                   const uint16_t *__restrict__ H, const uint16_t *__restrict__ Hinv,
                                                   uint8_t *__restrict__ tap_demap, uint8_t *__restrict__ tap_symdeint,
                                                   uint8_t *__restrict__ out, uint8_t *__restrict__ out_lp = nullptr)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint8_t *v = smem_raw;                                       // labels, padded block layout
  float2 *pts = reinterpret_cast<float2 *>(smem_raw + ((((p.payload + IB - 1) / IB) * IBS + 15) & ~15));
  unsigned char *label_of = reinterpret_cast<unsigned char *>(pts + 64);
  const int u = blockIdx.x, tid = threadIdx.x;
  constexpr int mode = MODE;                                    // compile time: no branches around the loads below
  int first = 0, nout = nitems_fixed;
  if (st) { first = st->first_out; nout = st->n_out_symbols; if (first < 0) return; }
  if (u >= nout) return;
  const int s = first + u;
  if (mode & 1) {
    for (int j = tid; j < p.csize; j += INNER_THREADS) pts[j] = points[j];
    if (tid < 64) label_of[tid] = label_tab[tid];
  }
  __syncthreads();
  const bool odd = (mode & 2) ? (sym_index[s] & 1) : false;
  const float2 *e = eq ? eq + (size_t)s * p.payload : nullptr;
  const uint8_t *ib = in_bytes ? in_bytes + (size_t)s * p.payload : nullptr;
  // batches of INNER_NB carriers per thread (a whole 8k symbol is one batch): every table/label load of a batch is
  // issued before the first LDS store, so a workgroup pays the memory latency once
  const uint16_t *ptab = odd ? H : Hinv;
  for (int q0 = tid; q0 < p.payload; q0 += INNER_NB * INNER_THREADS) {
    unsigned pk[INNER_NB];                                       // destination | label << 16
#pragma unroll
    for (int k = 0; k < INNER_NB; k++) {
      const int q = q0 + k * INNER_THREADS, qc = q < p.payload ? q : p.payload - 1;    // clamped: unconditional loads
      pk[k] = (mode & 2) ? (unsigned)ptab[qc] : (unsigned)qc;
      if (!(mode & 1)) pk[k] |= (unsigned)ib[qc] << 16;
    }
#pragma unroll
    for (int k = 0; k < INNER_NB; k++) {
      const int q = q0 + k * INNER_THREADS;
      if (q < p.payload) {
        const int lab = (mode & 1) ? demap_one(e[q], pts, label_of, p) : (int)(pk[k] >> 16);
        const int dst = (int)(pk[k] & 0xffffu);
        const int blk = dst / IB, r = dst - blk * IB, at = blk * IBS + r;
        v[at] = (uint8_t)lab;
        if (r < IBS - IB) v[at + IB] = (uint8_t)lab;             // the block's wrap-around tail
        if (tap_demap && (mode & 1)) tap_demap[(size_t)u * p.payload + q] = (uint8_t)lab;
      }
    }
  }
  __syncthreads();
  uint8_t *o = out + (size_t)u * p.payload;
  if (tap_symdeint) for (int q = tid; q < p.payload; q += INNER_THREADS) tap_symdeint[(size_t)u * p.payload + q] = v[ipos(q)];
  if (!(mode & 4)) { for (int q = tid; q < p.payload; q += INNER_THREADS) o[q] = v[ipos(q)]; return; }
  // A6: output byte i of a block, bit k (MSB first) = bit (m-1-e) of input byte (i - off_e) mod 126, e = perm(k).
  // One item = 4 consecutive output bytes: per bit plane one (unaligned) word of the rotated block, masked and
  // shifted into place.
  if (p.hier) bit_deint_hier(v, o, out_lp ? out_lp + (size_t)u * p.payload : nullptr, p.payload, p.m, tid);
  else if (p.m == 2) bit_deint_words<2>(v, o, p.payload, tid);
  else if (p.m == 4) bit_deint_words<4>(v, o, p.payload, tid);
  else bit_deint_words<6>(v, o, p.payload, tid);
}

// ---------------------------------------------------------------- sizes derived on the device (no host sync)
struct VitParams {
  int m, k, n, plen, ntb, bsize;
  int d_nsymbols;        // input bytes per reference block  (viterbi_decoder_impl.cc:149)
  int d_nbits;           // depunctured bits per block       (:151)
  int chunk_bytes;       // decoded bytes per wavefront chunk
  int payload;
  unsigned punct_mask;            // bit p = puncture vector entry p
  int warm;                       // warm-up windows in front of a chunk (read by the WARM = 0 instantiation of viterbi3_kernel only; sits in what was padding)
  unsigned long long prefix_nib;  // nibble p = kept bits before phase p
  unsigned long long magic_plen, magic_m;   // ceil(2^64/d): x/d == umul64hi(x, magic) for x < 2^56
  unsigned long long punct_rep;             // the puncture vector repeated over 64 bits
  unsigned magic16_plen;                    // ceil(2^16/plen): x/plen == (x*magic)>>16 for x < 1024
  uint8_t punct[16], prefix[16];
};

// sym_off = 0: the segment holds the beginning of a stream, every size is the reference's for a chain that starts at the
// segment's superframe start.  sym_off > 0 (a multiple of 272): the segment CONTINUES a cut stream whose
// first superframe start lies sym_off OFDM symbols before this segment's; the Viterbi block count
// (viterbi_decoder_impl.cc:198), the delay of ntraceback bytes and the even item count of the byte de-interleaver
// (set_output_multiple(2)) are then taken in STREAM coordinates, so that the segments' outputs end exactly where
// the outputs of one chain over the whole stream would.
__device__ __forceinline__ void plan_body(RxState *st, const VitParams &vp, long long sym_off)
{
  const long long ibits = (long long)vp.payload * vp.m * vp.k / vp.n;          // decoded bits per OFDM symbol
  const long long nin_g = (sym_off + (long long)st->n_out_symbols) * vp.payload;
  const long long nblocks_g = nin_g / vp.d_nsymbols;
  long long nin = nblocks_g * vp.d_nsymbols - sym_off * vp.payload;
  long long steps = nblocks_g * (vp.d_nbits / 2) - sym_off * ibits;
  if (nin < 0 || steps < 0) { nin = 0; steps = 0; }
  st->n_vit_in = nin;
  st->n_vit_steps = steps;
  long long nb_g = nblocks_g * (vp.d_nbits / 2) / 8 - vp.ntb;
  if (nb_g < 0) nb_g = 0;
  long long nb = steps / 8 - vp.ntb;
  if (nb < 0) nb = 0;
  st->n_vit_bytes = nb;
  const long long items_g = (nb_g / 1632) & ~1ll;               // convolutional_deinterleaver set_output_multiple(2)
  long long words = items_g * 8 - sym_off * ibits / (8 * 204);
  if (words < 0) words = 0;
  st->stream_rs_items = items_g;
  st->n_rs_words = words;
  st->n_rs_items = words / 8;
  st->sym_off = sym_off;
  st->n_ts_bytes = 0; st->rs_fail = 0; st->rs_corr = 0; st->ts_first_packet = 0; st->rs_list_n = 0;
}
// The tail of the TPS bookkeeping in one launch of one workgroup (three launches of ~5 us each before): the segment-parallel pass's neighbours compared
// (edges != nullptr; tps_finalize_body), the sequential bookkeeping when they disagree -- or always, for a lock period that carries the pilot engine's members on
// (edges == nullptr) --, then the sizes behind it (plan_body).
__global__ __launch_bounds__(256) void tps_tail_kernel(FrontParams p, RxState *st, const SymInfo *info, const int *maj, TpsState *ts, int *sym_index,
                                                      const TpsEdge *edges, const int *first_cand, int *need_seq, int clear_d_init, VitParams vp, long long sym_off)
{
  int seq = 1;
  if (edges) seq = tps_finalize_body(st, edges, first_cand, need_seq, p.keep_last, ts);
  if (seq) tps_fsm_body(p, st, 0, info, maj, ts, sym_index, (int *)nullptr, (const unsigned char *)nullptr, clear_d_init);
  if (threadIdx.x == 0) plan_body(st, vp, sym_off);              // thread 0 wrote n_out_symbols itself in either branch
}

// ---------------------------------------------------------------- A8+A9 fused: Forney de-interleave gather + RS(204,188) decode
// out word w, byte p  <-  viterbi stream word (w - 11 + p%12), byte p   (zeros before the start)
// (convolutional_deinterleaver_impl.cc:64-65,133-138 in closed form).  64 words per workgroup; the
// 75 source words are staged in LDS with coalesced loads, each thread then owns one codeword.
// RS: reed_solomon.cc:246-489.  Syndromes by table T_i[b] = b*alpha^i; error path on exp/log tables.
struct RsTables { const uint8_t *div_tab; /* [256][16]: b * g(x) */ const uint8_t *gexp; /* 512 */ const uint8_t *glog; /* 256 */ };

// Error path of rs_decode (reed_solomon.cc:315-486) for ONE codeword, executed by the whole
// wavefront: Berlekamp-Massey with one lane per polynomial coefficient, Chien search with four
// trial positions per lane, Forney with one lane per root.  Same results as the sequential code,
// including its order-dependent behaviours: roots are taken in ascending position and the search
// stops at deg(sigma) roots (:376-403); a zero Forney denominator aborts after the higher-indexed
// roots have already been patched (:445-486); compat reproduces the as-compiled omega[2t] overflow
// that redirects the lowest root's correction into the zero padding (SURVEY B-1).
// syn: 16 syndromes (LDS, RS_SYN_STRIDE bytes apart: syndrome i of the workgroup's word w is s_syn[i * RS_SYN_STRIDE + w]: an odd stride, so that
// the lanes of one word's Berlekamp-Massey and the words of one syndrome index both fall on distinct banks), cw: the 204 received bytes (LDS, index 0 = codeword index 51),
// scr: >= 64 bytes of LDS scratch.  Returns rs_decode's return value in every lane.
constexpr int RS_SYN_STRIDE = 65;
__device__ inline int rs_decode_word_wave(uint8_t *cw, const uint8_t *syn, const uint8_t *gexp, const uint8_t *glog,
                                          int compat, uint8_t *scr, int lane)
{
  auto gmul = [&](int a, int b) -> int { return (a == 0 || b == 0) ? 0 : gexp[glog[a] + glog[b]]; };
  auto gdiv = [&](int a, int b) -> int { return (a == 0 || b == 0) ? 0 : gexp[255 + glog[a] - glog[b]]; };
  auto wxor = [&](int v) -> int { for (int o = 16; o > 0; o >>= 1) v ^= __shfl_xor(v, o); return v; };   // over lanes 0..31
  // ---- Berlekamp-Massey: lane i (0..16) holds sigma[i] and b[i]
  int sig = lane == 0 ? 1 : 0, bb = sig, el = 0;
  for (int r = 1; r <= 16; r++) {
    int term = (lane < r && lane <= 16) ? gmul(sig, syn[(r - 1 - lane) * RS_SYN_STRIDE]) : 0;
    const int discr = __shfl(wxor(term), 0);
    int bup = __shfl_up(bb, 1); if (lane == 0) bup = 0;
    if (discr == 0) bb = bup;
    else {
      const int T = sig ^ gmul(discr, bup);
      if (2 * el <= r - 1) { el = r - el; bb = gdiv(sig, discr); } else bb = bup;
      sig = lane <= 16 ? T : 0;
    }
  }
  const unsigned long long nz = __ballot(sig != 0 && lane <= 16);
  const int deg_sigma = nz ? 63 - __clzll(nz) : 0;
  uint8_t *s_sig = scr, *s_root = scr + 17, *s_om = scr + 34;
  if (lane <= 16) s_sig[lane] = (uint8_t)sig;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  // ---- Chien: q(i) = 1 + sum_j sigma[j] alpha^(j*i), i = 1..255; lane tests i = lane+1 + 64k
  int found[4]; int nroots = 0;
  unsigned long long bal[4];
  for (int k = 0; k < 4; k++) {
    const int i = lane + 1 + 64 * k;
    int q = 1;
    if (i <= 255) {
      for (int j = 1; j <= deg_sigma; j++) { int sg = s_sig[j]; if (sg) q ^= gexp[(glog[sg] + (j * i) % 255) % 255]; }
    }
    found[k] = (i <= 255) && (q == 0);
    bal[k] = __ballot(found[k]);
  }
  {
    int before = 0;
    for (int k = 0; k < 4; k++) {
      if (found[k]) { int rank = before + __popcll(bal[k] & ((1ull << lane) - 1)); if (rank < 17 && rank < deg_sigma) s_root[rank] = (uint8_t)(lane + 1 + 64 * k); }
      before += __popcll(bal[k]);
    }
    nroots = before < deg_sigma ? before : deg_sigma;             // the reference stops searching at deg_sigma roots
  }
  if (nroots != deg_sigma) return -1;
  // ---- omega = sigma * S mod x^16 (lane i computes omega[i])
  {
    int tmp = 0;
    if (lane < 16) { int jm = deg_sigma < lane ? deg_sigma : lane; for (int j = jm; j >= 0; j--) tmp ^= gmul(syn[(lane - j) * RS_SYN_STRIDE], s_sig[j]); s_om[lane] = (uint8_t)tmp; }
    const unsigned long long oz = __ballot(lane < 16 && tmp != 0);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const int deg_omega = oz ? 63 - __clzll(oz) : 0;
    // ---- Forney: lane k handles root k
    int err = 0, den = 1, loc = 0;
    if (lane < nroots) {
      const int root = s_root[lane];
      int num1 = 0;
      for (int i = deg_omega; i >= 0; i--) { int om = s_om[i]; if (om) num1 ^= gexp[(glog[om] + (i * root) % 255) % 255]; }
      const int num2 = gexp[(255 - root) % 255];
      den = 0;
      const int deg_max = deg_sigma < 15 ? deg_sigma : 15;
      for (int i = 1; i <= deg_max; i += 2) { int sg = s_sig[i]; if (sg) den ^= gexp[(glog[sg] + ((i - 1) * root) % 255) % 255]; }
      err = gdiv(gmul(num1, num2), den);
      loc = (compat && lane == 0) ? 0 : root - 1;
    }
    const unsigned long long dz = __ballot(lane < nroots && den == 0);
    const int kz = dz ? 63 - __clzll(dz) : -1;                    // the reference walks roots from the highest index down
    if (lane < nroots && lane > kz && loc >= 51) cw[loc - 51] ^= (uint8_t)err;   // patches inside the zero padding are not output
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    return kz >= 0 ? -1 : nroots;
  }
}


// The same decoder with one WORD PER LANE (rs_decode's steps restated for 64 independent words; every lane walks its own word sequentially, the
// loops run to the wavefront's largest bound).  The wave-per-word version above costs ~6,000 issue slots of mostly waiting per bad word (dependent
// LDS look-ups with one wavefront per SIMD): fine for the odd bad word of a clean stream, a collapse when most words carry errors (8 dB QPSK 7/8:
// 3.7 ms per 344k words against 0.07 ms clean, tools/rs_load.py).  Here a wavefront's 64 words cost ~20,000 instructions together.  `active`: the
// lane has a word with a non-zero syndrome.  syn / sigma / omega live in registers (all loops over them are unrolled), the roots in a private LDS
// column (s_root[k * 64 + lane]).  Returns rs_decode's value (-1: uncorrectable, else the number of corrected symbols); inactive lanes return 0.
__device__ inline int rs_decode_word_lane(uint8_t *cw, const uint8_t *syn_col /* s_syn + lane */, const uint8_t *gexp, const uint8_t *glog, int compat,
                                          uint8_t *root_col /* s_root + lane */, bool active)
{
  auto gmul = [&](int a, int b) -> int { return (a == 0 || b == 0) ? 0 : gexp[glog[a] + glog[b]]; };
  int syn[16];
#pragma unroll
  for (int i = 0; i < 16; i++) syn[i] = active ? syn_col[i * RS_SYN_STRIDE] : 0;
  // ---- Berlekamp-Massey (:315-354)
  int sg[17], bb[17];
#pragma unroll
  for (int i = 0; i < 17; i++) { sg[i] = i == 0 ? 1 : 0; bb[i] = sg[i]; }
  int el = 0;
#pragma unroll
  for (int r = 1; r <= 16; r++) {
    int discr = 0;
#pragma unroll
    for (int i = 0; i < r; i++) discr ^= gmul(sg[i], syn[r - i - 1]);
    if (__any(discr != 0)) {
      const int dl = glog[discr];                                  // used only where discr != 0
      const bool grow = discr != 0 && 2 * el <= r - 1;
      int T[17];
      T[0] = sg[0];
#pragma unroll
      for (int i = 0; i < 16; i++) T[i + 1] = sg[i + 1] ^ ((discr == 0 || bb[i] == 0) ? 0 : gexp[dl + glog[bb[i]]]);
#pragma unroll
      for (int i = 16; i >= 0; i--) {
        const int shifted = i == 0 ? 0 : bb[i - 1];
        const int divd = (sg[i] == 0 || discr == 0) ? 0 : gexp[255 + glog[sg[i]] - dl];
        bb[i] = grow ? divd : shifted;
      }
      if (grow) el = r - el;
#pragma unroll
      for (int i = 0; i < 17; i++) sg[i] = discr != 0 ? T[i] : sg[i];
    } else {
#pragma unroll
      for (int i = 16; i >= 1; i--) bb[i] = bb[i - 1];
      bb[0] = 0;
    }
  }
  int deg_sigma = 0;
#pragma unroll
  for (int i = 1; i < 17; i++) if (sg[i]) deg_sigma = i;
  if (!active) deg_sigma = 0;
  // ---- Chien (:376-403): roots in ascending position, the search stops at deg(sigma) roots.  Term j in the log domain: l_j = log sigma_j + j i (mod 255)
  int lj[17], mj[17];
#pragma unroll
  for (int j = 1; j < 17; j++) { lj[j] = sg[j] ? glog[sg[j]] : 0; mj[j] = (sg[j] && j <= deg_sigma) ? 0xff : 0; }
  int maxdeg = deg_sigma;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(maxdeg, o); maxdeg = v > maxdeg ? v : maxdeg; }
  int no_roots = 0;
  bool done = deg_sigma == 0;                                      // (an active word always has deg >= 1)
  for (int i = 1; i <= 255; i++) {
    int q = 1;
#pragma unroll
    for (int j = 1; j < 17; j++) {
      if (j <= maxdeg) {
        int t = lj[j] + j; t = t >= 255 ? t - 255 : t; lj[j] = t;
        q ^= gexp[t] & mj[j];
      }
    }
    if (!done && q == 0) {
      root_col[no_roots * 64] = (uint8_t)i;
      if (++no_roots == deg_sigma) done = true;
    }
    if (__all(done)) break;
  }
  bool failed = active && no_roots != deg_sigma;                   // :405-415
  // ---- omega = sigma S mod x^16 (:419-434)
  int om[16], deg_omega = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    int tmp = 0;
#pragma unroll
    for (int j = 0; j <= i; j++) tmp ^= (j <= deg_sigma) ? gmul(syn[i - j], sg[j]) : 0;
    om[i] = tmp; if (tmp) deg_omega = i;
  }
  int lo[16];
#pragma unroll
  for (int i = 0; i < 16; i++) lo[i] = om[i] ? glog[om[i]] : -1;
  int ls[16];                                                      // log sigma_i for odd i (the formal derivative, :466-470)
#pragma unroll
  for (int i = 1; i < 16; i += 2) ls[i] = sg[i] ? glog[sg[i]] : -1;
  // ---- Forney (:445-486): roots from the highest index down; a zero denominator aborts after the higher roots have been patched
  int nr = (active && !failed) ? no_roots : 0, maxr = nr;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(maxr, o); maxr = v > maxr ? v : maxr; }
  for (int jj = maxr - 1; jj >= 0; jj--) {
    const bool on = jj < nr && !failed;
    const int root = on ? root_col[jj * 64] : 1;
    int num1 = 0, den = 0, ir = 0;                                  // ir = i * root mod 255
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (i <= deg_omega && lo[i] >= 0) { const int e = lo[i] + ir; num1 ^= gexp[e >= 255 ? e - 255 : e]; }
      if ((i & 1) == 0 && i + 1 < 16 && i + 1 <= deg_sigma && ls[i + 1] >= 0) { const int e = ls[i + 1] + ir; den ^= gexp[e >= 255 ? e - 255 : e]; }   // (i + 1 - 1) * root
      ir += root; ir = ir >= 255 ? ir - 255 : ir;
    }
    if (on) {
      if (den == 0) failed = true;
      else {
        const int num2 = gexp[(255 - root) % 255];
        const int err = (num1 == 0) ? 0 : gexp[255 + glog[gmul(num1, num2)] - glog[den]];
        const int loc = (compat && jj == 0) ? 0 : root - 1;
        if (loc >= 51) cw[loc - 51] ^= (uint8_t)err;               // patches inside the zero padding are not output
      }
    }
  }
  if (!active) return 0;
  return failed ? -1 : no_roots;
}

// A wavefront with only a few bad words (a clean stream's junction words, the odd error burst) hands them to a second launch, where every bad word gets
// a wavefront of its own: the 11 undecodable start-up words of a stream (convolutional_deinterleaver_impl.cc:64-65) sit in ONE workgroup and were the
// critical path of the whole launch (11 x ~8 us in sequence while the other 5,000 workgroups had long finished)
struct RsDefer { int word; uint8_t syn[16]; uint8_t cw[204]; };   // 224 bytes: the word's index, its syndromes, the received codeword
constexpr int RS_LANE_MIN = 24;                                // bad words per wavefront from which every lane decodes its own word
// standalone = 1: input is already de-interleaved items (A9 block alone); 0: gather from the Viterbi stream (A8+A9)
__global__ __launch_bounds__(64) void deint_rs_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ deint_tap,
                                                     uint8_t *__restrict__ out, RxState *st, long long words_fixed, int standalone,
                                                     long long hist_words /* words of real history before word 0 (block API) */,
                                                     RsTables T, int compat, int *fail_cnt, int *corr_cnt,
                                                     RsDefer *__restrict__ defer = nullptr, int *defer_n = nullptr, int defer_cap = 0,
                                                     unsigned long long *__restrict__ sync_bits = nullptr /* bit w: payload byte 0 of word w is the inverted sync byte 0xB8 (descramble_scan_kernel) */)
{
  // codewords of the workgroup, one row of 204 bytes each; 11 spare rows on either side absorb the bytes of the 75
  // source words that belong to codewords of the neighbouring workgroups, so the scatter below needs no bounds test
  __shared__ __attribute__((aligned(16))) uint8_t s_rows[(64 + 22) * 204];
  __shared__ __attribute__((aligned(16))) uint8_t s_div[256 * 16];
  __shared__ uint8_t s_exp[512], s_log[256];
  __shared__ uint8_t s_scr[64];
  __shared__ uint8_t s_syn[RS_SYN_STRIDE * 16];
  __shared__ uint8_t s_root[17 * 64];                          // rs_decode_word_lane: the roots found, one column per lane
  uint8_t *s_cw = s_rows + 11 * 204;
  const int tid = threadIdx.x;
  const long long nwords = st ? st->n_rs_words : words_fixed;
  const long long w0 = (long long)blockIdx.x * 64;
  if (w0 >= nwords) return;
  for (int i = tid; i < 1024; i += 64) reinterpret_cast<unsigned *>(s_div)[i] = reinterpret_cast<const unsigned *>(T.div_tab)[i];
  for (int i = tid; i < 512; i += 64) s_exp[i] = T.gexp[i];
  for (int i = tid; i < 256; i += 64) s_log[i] = T.glog[i];
  const int nw = (int)((nwords - w0) < 64 ? (nwords - w0) : 64);
  if (!standalone) {
    // A8 (convolutional_deinterleaver_impl.cc:133-138 in closed form): byte p of codeword w is byte p of source word
    // w - 11 + p % 12.  The 75 source words w0-11 .. w0+63 are read as dwords (204 = 51 dwords, 4-byte aligned) and
    // every byte goes straight to its codeword's row: byte p of source row r belongs to row r - 11 - ... = r - p % 12
    // (rows counted from w0 - 11), i.e. the four bytes of a dword land 203 bytes apart.
    const long long first = w0 - 11;
    for (int i0 = tid; i0 < 75 * 51; i0 += 6 * 64) {               // 6 dword loads in flight per lane (60 = 10 x 6 per lane)
      unsigned v[6];
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const int i = i0 + k * 64, r = i / 51, p4 = i - r * 51;
        const long long word = first + r;
        const bool valid = i < 75 * 51 && word >= -hist_words && word < nwords && r < nw + 11;
        const long long wc = valid ? word : (nwords > 0 ? 0 : -hist_words);      // unconditional load from a valid address, selected afterwards
        const unsigned ld = reinterpret_cast<const unsigned *>(in + wc * 204)[valid ? p4 : 0];
        v[k] = valid ? ld : 0u;
      }
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const int i = i0 + k * 64, r = i / 51, p4 = i - r * 51;
        if (i < 75 * 51) {
          uint8_t *d = s_rows + (r + 11 - (p4 % 3) * 4) * 204 + 4 * p4;         // row of byte 0 of this dword (s_rows row = codeword - w0 + 11)
          d[0] = (uint8_t)v[k]; d[1 - 204] = (uint8_t)(v[k] >> 8); d[2 - 408] = (uint8_t)(v[k] >> 16); d[3 - 612] = (uint8_t)(v[k] >> 24);
        }
      }
    }
  } else {
    for (int i = tid; i < nw * 51; i += 64) reinterpret_cast<unsigned *>(s_cw)[i] = reinterpret_cast<const unsigned *>(in + w0 * 204)[i];
  }
  __syncthreads();
  if (deint_tap) for (int i = tid; i < nw * 51; i += 64) reinterpret_cast<unsigned *>(deint_tap + w0 * 204)[i] = reinterpret_cast<const unsigned *>(s_cw)[i];
  const uint8_t *cw = s_cw + tid * 204;
  bool bad = false;
  if (tid < nw) {
    // remainder of the received word modulo g(x): R <- R*x + c (mod g), one 16-byte table row per byte.  All 16
    // syndromes S_i = C(alpha^i) = R(alpha^i) vanish iff R == 0 (reed_solomon.cc:281-305), so clean words stop here.
    unsigned R0 = 0, R1 = 0, R2 = 0, R3 = 0;                       // R0 byte 0 = coefficient of x^0 ... R3 byte 3 = x^15
    for (int p4 = 0; p4 < 51; p4++) {
      const unsigned d = reinterpret_cast<const unsigned *>(cw)[p4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint4 t = *reinterpret_cast<const uint4 *>(s_div + (R3 >> 24) * 16);
        R3 = ((R3 << 8) | (R2 >> 24)) ^ t.w; R2 = ((R2 << 8) | (R1 >> 24)) ^ t.z;
        R1 = ((R1 << 8) | (R0 >> 24)) ^ t.y; R0 = ((R0 << 8) | ((d >> (8 * k)) & 0xffu)) ^ t.x;
      }
    }
    // (Four bytes per step on four tables of four-step rows -- one dependent LDS round trip per dword instead of four -- was built and measured in round 4:
    // 117 us against 92: the 16 KB of tables leave room for four workgroups per compute unit instead of six, and the launch is bound by how many words a
    // compute unit has in flight, not by the chain.)
    const int any = (R0 | R1 | R2 | R3) != 0;
    if (any) {                                                    // syndromes from the remainder: S_i = sum_k R_k alpha^(i k)
      const unsigned Rw[4] = {R0, R1, R2, R3};
      for (int i = 0; i < 16; i++) {
        int sv = 0;
        for (int k = 15; k >= 0; k--) { int c = (Rw[k >> 2] >> (8 * (k & 3))) & 0xff; sv = (sv ? s_exp[(s_log[sv] + i) % 255] : 0) ^ c; }
        s_syn[i * RS_SYN_STRIDE + tid] = (uint8_t)sv;
      }
    }
    bad = any != 0;
  }
  // words with a non-zero syndrome (reed_solomon.cc:299-305 returns early otherwise): a few of them are decoded one at a time by the whole
  // wavefront (lowest latency), more than RS_LANE_MIN by every lane for itself (rs_decode_word_lane: ~18x the throughput when most words are bad)
  {
    unsigned long long mask = __ballot(bad);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    int nf = 0, nc = 0;
    if (__popcll(mask) >= RS_LANE_MIN) {
      const int r = rs_decode_word_lane(s_cw + tid * 204, s_syn + tid, s_exp, s_log, compat, s_root + tid, bad);
      const unsigned long long fm = __ballot(bad && r < 0);
      nf = __popcll(fm);
      int c = (bad && r > 0) ? r : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
      nc = c;
    } else if (defer && mask) {
      // second pass (rs_fix_kernel): the word's index, syndromes and received bytes go to the list; its payload is stored below as received
      const int cnt = __popcll(mask);
      int base = 0;
      if (tid == 0) base = atomicAdd(defer_n, cnt);
      base = __shfl(base, 0);
      if (bad) {
        const int e = base + __popcll(mask & ((1ull << tid) - 1));
        if (e < defer_cap) {
          RsDefer *d = defer + e;
          d->word = (int)(w0 + tid);
#pragma unroll
          for (int i = 0; i < 16; i++) d->syn[i] = s_syn[i * RS_SYN_STRIDE + tid];
          for (int i = 0; i < 51; i++) reinterpret_cast<unsigned *>(d->cw)[i] = reinterpret_cast<const unsigned *>(cw)[i];
        }
      }
    } else {
      while (mask) {
        const int wl = __ffsll((long long)mask) - 1; mask &= mask - 1;
        const int r = rs_decode_word_wave(s_cw + wl * 204, s_syn + wl, s_exp, s_log, compat, s_scr, tid);
        if (r < 0) nf++; else nc += r;
      }
    }
    if (tid == 0) { if (nf) atomicAdd(fail_cnt, nf); if (nc) atomicAdd(corr_cnt, nc); }
  }
  __syncthreads();
  if (sync_bits) { const unsigned long long m = __ballot(tid < nw && s_cw[tid * 204] == 0xB8); if (tid == 0) sync_bits[blockIdx.x] = m; }
  // coalesced store of 64 x 188 payload bytes as dwords (reed_solomon_dec_impl.cc:102: output regardless of success)
  unsigned *o4 = reinterpret_cast<unsigned *>(out + w0 * 188);
  for (int i = tid; i < nw * 47; i += 64) { const int ww = i / 47, q = i - ww * 47; o4[i] = reinterpret_cast<const unsigned *>(s_cw + ww * 204)[q]; }
}

// second pass of A9: one wavefront per listed word (rs_decode_word_wave), the corrected payload written over the uncorrected one
__global__ __launch_bounds__(64) void rs_fix_kernel(const RsDefer *__restrict__ list, const int *__restrict__ list_n, int list_cap, uint8_t *__restrict__ out,
                                                   RsTables T, int compat, int *fail_cnt, int *corr_cnt, unsigned long long *sync_bits = nullptr)
{
  __shared__ __attribute__((aligned(16))) uint8_t s_cw[208];
  __shared__ uint8_t s_exp[512], s_log[256], s_scr[64], s_syn[RS_SYN_STRIDE * 16];
  int n = *list_n; if (n > list_cap) n = list_cap;
  if ((int)blockIdx.x >= n) return;
  const int tid = threadIdx.x;
  for (int i = tid; i < 512; i += 64) s_exp[i] = T.gexp[i];
  for (int i = tid; i < 256; i += 64) s_log[i] = T.glog[i];
  int nf = 0, nc = 0;
  for (int e = blockIdx.x; e < n; e += gridDim.x) {
    const RsDefer *d = list + e;
    __syncthreads();
    if (tid < 51) reinterpret_cast<unsigned *>(s_cw)[tid] = reinterpret_cast<const unsigned *>(d->cw)[tid];
    if (tid < 16) s_syn[tid * RS_SYN_STRIDE] = d->syn[tid];
    __syncthreads();
    const int r = rs_decode_word_wave(s_cw, s_syn, s_exp, s_log, compat, s_scr, tid);
    if (r < 0) nf++; else nc += r;
    // reed_solomon_dec_impl.cc:100-102: the payload leaves regardless of success, with the patches applied before a zero Forney denominator aborted
    // (reed_solomon.cc:470-486) -- as the first pass's wave and lane paths deliver it
    if (tid < 47) reinterpret_cast<unsigned *>(out + (size_t)d->word * 188)[tid] = reinterpret_cast<const unsigned *>(s_cw)[tid];
    if (sync_bits && tid == 0) {                                 // the first pass recorded the word's sync bit as received
      const unsigned long long m = 1ull << (d->word & 63);
      if (s_cw[0] == 0xB8) atomicOr(&sync_bits[d->word >> 6], m); else atomicAnd(&sync_bits[d->word >> 6], ~m);
    }
  }
  if (tid == 0) { if (nf) atomicAdd(fail_cnt, nf); if (nc) atomicAdd(corr_cnt, nc); }
}

// A8 alone (block API): buf holds `hist` bytes of history followed by the call's n input bytes;
// stream index of buf[hist] is `pos0`.  out[i] = stream[pos0+i - 204*(11-(pos0+i)%12)], 0 before the start.
__global__ __launch_bounds__(256) void conv_deint_kernel(const uint8_t *__restrict__ buf, long long hist, long long pos0, long long n,
                                                        uint8_t *__restrict__ out)
{
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    long long g = pos0 + i, d = 204ll * (11 - (g % 12)), src = g - d;
    out[i] = (src >= 0 && i - d + hist >= 0) ? buf[hist + i - d] : 0;
  }
}

// ---------------------------------------------------------------- next row: energy_descramble (energy_descramble_impl.cc:108-174)
// The block is restated call by call in the smallest calls it accepts (4 items visible, 2 consumed and delivered: :90,:139-141).
// Every call first looks at the byte at its offset d_index: no NSYNC (0xB8) there -> search on at 188-byte strides within the first
// two items (:121-123); nothing found -> offset back to 0, two items dropped, nothing delivered (:129-134).  One workgroup walks the
// calls: all lanes test the bytes of the coming calls for the current offset at once, the first mismatch (if any) is handled by one
// lane exactly as above.  What is delivered is a list of runs of whole items (normally one: lock on the first NSYNC, deliver from
// there to two items before the end); descramble_runs_kernel copies them through the PRBS.
struct DescrRun { long long src_byte, dst_byte, nbytes; };
constexpr int DESCR_MAX_RUNS = 1024;

// sync_bits (the segment chain): bit w = payload byte 0 of RS word w is 0xB8, written by deint_rs_kernel / rs_fix_kernel -- the walk then reads one bit per
// packet out of a few KB instead of one byte per 188 out of the whole TS (21,000 cache lines through ONE compute unit: 45 us of the 65-superframe step);
// nullptr: the bytes themselves.
__global__ __launch_bounds__(1024) void descramble_scan_kernel(const uint8_t *__restrict__ in, RxState *st, DescrRun *runs, int *nruns,
                                                              const unsigned long long *__restrict__ sync_bits = nullptr, int phase16 = -1)
{
  __shared__ long long s_first;
  __shared__ long long s_base, s_written; __shared__ int s_dindex, s_nr, s_stop, s_bad;
  const int tid = threadIdx.x;
  auto nsync = [&](long long pkt) -> bool {                        // is the first byte of packet `pkt` the inverted sync byte?
    return sync_bits ? ((sync_bits[pkt >> 6] >> (pkt & 63)) & 1ull) != 0 : in[pkt * 188] == 0xB8;
  };
  if (tid == 0) { st->n_ts_bytes = 0; st->descr_base = 0; st->descr_index = 0; st->ts_first_packet = 0; st->descr_unclean = 0; *nruns = 0; }
  if (st->sym_off > 0) {
    // continuation of a cut stream: the descrambler of the whole-stream chain locked long ago; this segment delivers
    // every whole 8-packet group from its first NSYNC on (the words before it mix the de-interleaver's zero fill with
    // data, exactly like the first words of a stream; their sync positions hold zeros).  Which of them the stitched
    // stream keeps is the host's decision (gr_dvbt_amd/multi.py::stitch_ts): the two-item hold-back of :139-141
    // belongs to the stream's end, not to a cut.
    // phase16 >= 0: the whole-stream descrambler runs in calls of two items whose first packets are == phase16 (mod 16) among this segment's RS words (the
    // streaming entry knows that phase): every such call that lies inside the segment's valid words must find its NSYNC, else the stream's descrambler
    // searches again here and "every whole group" is not what it delivers -- reported (descr_unclean), the host then walks the calls itself
    const long long nw = st->n_rs_words;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    if (phase16 >= 0) {
      long long s0 = phase16 & 15; while (s0 < 11) s0 += 16;
      int bad = 0;
      for (long long c = s0 + 16ll * tid; c + 32 <= nw; c += 16ll * 1024) if (!nsync(c)) bad = 1;
      if (bad) s_bad = 1;
    }
    __syncthreads();
    if (tid == 0) {
      long long q = 0;
      while (q < nw && !nsync(q)) q++;
      if (q < nw) {
        st->descr_index = (int)(q * 188); st->ts_first_packet = q;
        st->n_ts_bytes = ((nw - q) / 8) * 1504;
        runs[0].src_byte = q * 188; runs[0].dst_byte = 0; runs[0].nbytes = st->n_ts_bytes; *nruns = st->n_ts_bytes > 0 ? 1 : 0;
      }
      st->descr_unclean = s_bad;
    }
    return;
  }
  const long long nitems = st->n_rs_items;
  if (tid == 0) { s_base = 0; s_written = 0; s_dindex = 0; s_nr = 0; s_stop = 0; }
  __syncthreads();
  while (true) {
    const long long base = s_base; const int d_index = s_dindex, d_pkt = d_index / 188;
    const long long ncalls = nitems - base >= 4 ? (nitems - base - 4) / 2 + 1 : 0;      // calls that still see 4 items
    if (ncalls == 0 || s_stop) break;
    if (tid == 0) s_first = ncalls;
    __syncthreads();
    long long mine = ncalls;
    // the call at `base` itself first (every lane reads the same bit): without its NSYNC there is nothing to scan for -- the first round of every stream,
    // whose offset 0 is not a packet start of the descrambler's phase
    if (!nsync(base * 8 + d_pkt)) mine = 0;
    for (long long k0 = tid; k0 < ncalls && mine == ncalls; k0 += 8 * 1024) {       // 8 clamped loads per lane in flight
      bool v[8];
#pragma unroll
      for (int j = 0; j < 8; j++) { const long long k = k0 + j * 1024; v[j] = nsync((base + 2 * (k < ncalls ? k : ncalls - 1)) * 8 + d_pkt); }
#pragma unroll
      for (int j = 7; j >= 0; j--) { const long long k = k0 + j * 1024; if (k < ncalls && !v[j]) mine = k; }
    }
    if (mine < ncalls && (mine > 0 || tid == 0)) atomicMin((unsigned long long *)&s_first, (unsigned long long)mine);
    __syncthreads();
    if (tid == 0) {
      const long long kf = s_first;
      if (kf > 0) {                                                // kf calls in a row found their NSYNC: one run
        const int r = s_nr;
        if (r < DESCR_MAX_RUNS) {
          if (r == 0) { st->descr_base = (int)base; st->descr_index = d_index; st->ts_first_packet = (base * 1504 + d_index) / 188; }
          runs[r].src_byte = base * 1504 + d_index; runs[r].dst_byte = s_written; runs[r].nbytes = kf * 2 * 1504; s_nr = r + 1;
          s_written += kf * 2 * 1504;
        } else s_stop = 1;
        s_base = base + 2 * kf;
      }
      if (kf < ncalls) {                                           // the call at s_base: search on, or give up and drop two items
        const long long p0 = s_base * 8;
        int dp = d_pkt;
        while (dp < 16 && !nsync(p0 + dp)) dp++;
        if (dp >= 16) { dp = 0; s_base += 2; }
        s_dindex = dp * 188;
      }
    }
    __syncthreads();
  }
  if (tid == 0) { st->n_ts_bytes = s_written; *nruns = s_nr; }
}

// a dword per thread and pass over the delivered bytes (an 8-packet group is 1504 bytes = 376 dwords of the PRBS pattern; every run starts on a multiple
// of 188 bytes); a workgroup per group left a third of its lanes idle in the second of its two passes
__global__ __launch_bounds__(256) void descramble_runs_kernel(const uint8_t *__restrict__ in, const uint8_t *__restrict__ seq, const RxState *st,
                                                             const DescrRun *__restrict__ runs, const int *nruns, uint8_t *__restrict__ out)
{
  const unsigned long long ndw = (unsigned long long)(st->n_ts_bytes / 1504) * 376ull;
  const unsigned *sq = reinterpret_cast<const unsigned *>(seq);
  const int nr = *nruns;
  const long long src0 = nr > 0 ? runs[0].src_byte : 0;
#pragma unroll 4
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < ndw; i += (unsigned long long)gridDim.x * 256) {
    const unsigned long long g = i / 376ull; const unsigned t = (unsigned)(i - g * 376ull);
    const long long dst = (long long)g * 1504;
    long long src = src0 + dst;                                  // one run (the usual case): it starts at destination byte 0
    if (nr > 1) {
      int r = 0;
      while (r + 1 < nr && runs[r + 1].dst_byte <= dst) r++;
      src = runs[r].src_byte + (dst - runs[r].dst_byte);
    }
    unsigned v = reinterpret_cast<const unsigned *>(in + src)[t] ^ sq[t];
    if (t % 47u == 0u) v = (v & 0xffffff00u) | 0x47u;              // sync byte restored (:151)
    reinterpret_cast<unsigned *>(out + dst)[t] = v;
  }
}

// dword-wise: one workgroup pass per 8-packet group (1504 bytes = 376 dwords; the start offset is a multiple of 188)
__global__ __launch_bounds__(256) void descramble_apply_kernel(const uint8_t *__restrict__ in, const uint8_t *__restrict__ seq,
                                                              const RxState *st, uint8_t *__restrict__ out)
{
  const long long ngroups = st->n_ts_bytes / 1504;
  const unsigned *p = reinterpret_cast<const unsigned *>(in + (long long)st->descr_base * 1504 + st->descr_index);
  const unsigned *sq = reinterpret_cast<const unsigned *>(seq);
  unsigned *o = reinterpret_cast<unsigned *>(out);
  for (long long g = blockIdx.x; g < ngroups; g += gridDim.x)
    for (int t = threadIdx.x; t < 376; t += 256) {
      unsigned v = p[g * 376 + t] ^ sq[t];
      if (t % 47 == 0) v = (v & 0xffffff00u) | 0x47u;              // sync byte restored (:151)
      o[g * 376 + t] = v;
    }
}

}  // namespace dvbt
