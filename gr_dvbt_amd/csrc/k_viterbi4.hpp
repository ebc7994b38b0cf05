// k_viterbi4.hpp -- A7, fourth generation: the packed in-place DPP trellis of k_viterbi3.hpp with EIGHT chunks per
// wavefront.  A chunk owns 8 lanes (half a DPP row); a lane holds 8 of the 64 path metrics as the halves of four
// VGPRs.  Cell index
//     c = r1 r0 (VGPR) : h (half) : a2 a1 a0,      physical lane-in-group g = (a0 + 2*a1) ^ (7*a2)
// and cell c holds state rotl6(c, u mod 6) at relative step u.  The six exchanges are two register renamings (free),
// one half swap (v_alignbit) and three DPP controls (row_half_mirror, quad_perm[2,3,0,1], quad_perm[1,0,3,2]).
// Against v3 (4 chunks x 16 lanes x 4 cells) the add-compare-select work per chunk is the same, but
//   * one third fewer exchange instructions and fewer v_perm (the four VGPRs share or negate each other's deltas:
//     1 or 2 v_perm per step instead of 4),
//   * the per-window work that does not depend on the number of chunks (best-state reduction, renormalisation,
//     table store, loop) is shared by 8 chunks instead of 4 and its DPP reductions are one level shorter.
// Cell format, tie rule, path bytes, traceback, depuncturing: exactly as k_viterbi3.hpp (see there).  The LDS ring is
// twice as large per wavefront (8 decoders), so only one wavefront is resident per SIMD: all latencies (input bytes,
// traceback hops, staging) are overlapped inside the wavefront (prefetch one block ahead, hops riding in the forward
// windows).  Blocks are 12 windows (96 steps: 12 steps per lane and decoder in the staging).
//
// MEASURED (MI355X, 8k QAM64 7/8, 65 superframes): 4.98 ms against 4.63 ms of viterbi3_kernel, although it executes ~20 %
// fewer VALU instructions per chunk: with a single wavefront per SIMD the scalar instructions, waits and DPP/LDS
// hazards of that wavefront are no longer hidden behind a second one.  Not the default (DVBT_VITERBI_KERNEL=4 selects
// it); it would pay off only with an LDS ring small enough for two wavefronts per SIMD.
#pragma once
#include "k_viterbi3.hpp"

namespace dvbt {

constexpr int V4_WARM = 72;        // warm-up windows before a chunk's first byte
constexpr int V4_BLK = 12;         // windows per block (multiple of 6)
constexpr int V4_RINGW = 64;       // windows in the LDS ring (>= 2*V4_BLK + max ntraceback - 1)
constexpr int V4_CBW = 12;         // words of compacted received bits per decoder and block (96 steps need <= 192 + 23 bits)
constexpr int V4_DEC = 8;          // decoders per wavefront

struct V4Lane {
  unsigned selA[6], selB[6];   // v_perm selectors of the (at most) two distinct delta registers of a step
  int abit[3];                 // logical lane bits a0..a2 at the bias position of both halves
  int hi_bias;                 // 0x01000000
  int kc[3][4];                // 63 - state of the two cells of VGPR r at window ends (phase 0,2,4)
  int nb2[3][4];               // bias of phase 0,2,4 | the two oldest inputs (stamp of the 6th step)
  int nbo[3][4];               // bias of phase 0,2,4 | origin stamp (last step)
  int org[4];
};

__device__ __forceinline__ int v4_cell_of_z(int z) { return (((z >> 1) & 3) << 4) | ((z & 1) << 3) | v3_log(z >> 3); }
__device__ __forceinline__ int v4_z_of_cell(int c) { return v3_phys(c & 7) * 8 + ((c >> 4) & 3) * 2 + ((c >> 3) & 1); }

__device__ inline void v4_init_lane(int g, V4Lane &L)
{
  const int a = v3_log(g);
  for (int k = 0; k < 3; k++) L.abit[k] = ((a >> k) & 1) * 0x01000100;
  L.hi_bias = vconst(0x01000000);
  auto sel = [&](int P, int r) {
    unsigned idx[2];
    for (int h = 0; h < 2; h++) {
      const int c = (r << 4) | (h << 3) | a, i = rotl6(c, P) & 31;
      const int c0 = ((i >> 2) ^ (i >> 1) ^ i) & 1;                   // parity(2i & 0x4f)
      const int c1 = ((i >> 4) ^ (i >> 2) ^ (i >> 1)) & 1;            // parity(2i & 0x6d)
      idx[h] = (unsigned)(c0 | (c1 << 1));
    }
    return 0x0cu | (idx[0] << 8) | (0x0cu << 16) | (idx[1] << 24);
  };
  for (int P = 0; P < 6; P++) { L.selA[P] = sel(P, 0); L.selB[P] = sel(P, (P == 0 || P == 2) ? 1 : 2); }
  for (int r = 0; r < 4; r++) {
    L.org[r] = ((g * 8 + 2 * r) << 2) | (((g * 8 + 2 * r + 1) << 2) << 16);
    for (int e = 0; e < 3; e++) {
      const int s0 = rotl6((r << 4) | a, 2 * e), s1 = rotl6((r << 4) | 8 | a, 2 * e);
      L.kc[e][r] = (63 - s0) | ((63 - s1) << 16);
      const int b2 = (s0 >> 4) | ((s1 >> 4) << 16);
      const int nb = e == 0 ? ((r & 2) ? 0x01000100 : 0) : e == 1 ? 0x01000000 : L.abit[1];   // bias when the phase is 0 (bit r1), 2 (half), 4 (a1)
      L.nb2[e][r] = nb | b2; L.nbo[e][r] = nb | L.org[r];
    }
  }
}

// one trellis step at phase P; ST as in v3_step
template <int P, int ST> __device__ __forceinline__ void v4_step(int (&v)[4], unsigned W, const V4Lane &L, int (&raw)[4])
{
  // Deltas.  The VGPR index bits r1 r0 are cell bits 5,4 = state bits (5+P)%6, (4+P)%6; the label class depends on state
  // bits 0,1,2,4 (flipping bit 1 or 2 negates the delta, bit 0 or 4 changes it, bit 3 or 5 leaves it):
  //   P0: r0 changes | P1: r1 changes | P2: r1 negates, r0 changes | P3: r1, r0 negate | P4: r0 negates | P5: r1 changes
  const int Da = (int)__builtin_amdgcn_perm(0u, W, L.selA[P]);
  const int Db = (P == 0 || P == 1 || P == 2 || P == 5) ? (int)__builtin_amdgcn_perm(0u, W, L.selB[P]) : 0;
  int X[4], Y[4], Yp[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int r0 = r & 1, r1 = r >> 1;
    int D; bool neg;
    if (P == 0) { D = r0 ? Db : Da; neg = false; }
    else if (P == 1 || P == 5) { D = r1 ? Db : Da; neg = false; }
    else if (P == 2) { D = r0 ? Db : Da; neg = r1; }
    else if (P == 3) { D = Da; neg = (r0 ^ r1) != 0; }
    else { D = Da; neg = r0; }
    X[r] = neg ? pk_sub(v[r], D) : pk_add(v[r], D);
    Y[r] = neg ? pk_add(v[r], D) : pk_sub(v[r], D);
  }
#pragma unroll
  for (int r = 0; r < 4; r++)
    Yp[r] = P == 0 ? Y[r ^ 2] : P == 1 ? Y[r ^ 1] : P == 2 ? swap16(Y[r]) : P == 3 ? dppb<DPP_HALF_MIRROR>(Y[r]) : P == 4 ? dppb<DPP_XOR2>(Y[r]) : dppb<DPP_XOR1>(Y[r]);
  constexpr int PN = (P + 1) % 6;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int mx = pk_max(X[r], Yp[r]);
    if (ST == 1) v[r] = (mx & (int)0xfefcfefc) | L.nb2[PN / 2][r];
    else if (ST == 2) { raw[r] = mx; v[r] = (mx & (int)0xfe00fe00) | L.nbo[PN / 2][r]; }
    else if (PN == 0) v[r] = (r & 2) ? (mx | 0x01000100) : (mx & (int)0xfefffeff);
    else if (PN == 1) v[r] = (r & 1) ? (mx | 0x01000100) : (mx & (int)0xfefffeff);
    else {
      const int nb = PN == 2 ? L.hi_bias : PN == 3 ? L.abit[2] : PN == 4 ? L.abit[1] : L.abit[0];
      v[r] = (mx & (int)0xfefffeff) | nb;
    }
  }
}

template <int P0> __device__ __forceinline__ void v4_window(int (&v)[4], const unsigned (&W)[8], const V4Lane &L, int (&raw)[4])
{
  v4_step<(P0 + 0) % 6, 0>(v, W[0], L, raw); v4_step<(P0 + 1) % 6, 0>(v, W[1], L, raw); v4_step<(P0 + 2) % 6, 0>(v, W[2], L, raw);
  v4_step<(P0 + 3) % 6, 0>(v, W[3], L, raw); v4_step<(P0 + 4) % 6, 0>(v, W[4], L, raw); v4_step<(P0 + 5) % 6, 1>(v, W[5], L, raw);
  v4_step<(P0 + 6) % 6, 0>(v, W[6], L, raw); v4_step<(P0 + 7) % 6, 2>(v, W[7], L, raw);
}

// best state of the window (first index of the maximum, d_viterbi.c:699-711) in every lane of the group; optional renormalisation
template <int PE, bool RENORM> __device__ __forceinline__ int v4_window_end(int (&v)[4], const V4Lane &L)
{
  const v3pk sixty4 = {64, 64};
  int kp[4];
#pragma unroll
  for (int r = 0; r < 4; r++) kp[r] = ipk((pk(v[r]) >> 9) * sixty4 + pk(L.kc[PE / 2][r]));
  const int kq = pk_max(pk_max(kp[0], kp[1]), pk_max(kp[2], kp[3]));
  int k = max(lo16(kq), hi16(kq));
  k = max_dpp<DPP_XOR1>(k); k = max_dpp<DPP_XOR2>(k); k = max_dpp<DPP_HALF_MIRROR>(k);
  if (RENORM) {
    const int mp = pk_min(pk_min(v[0], v[1]), pk_min(v[2], v[3]));
    int mn = min(lo16(mp), hi16(mp));
    mn = min_dpp<DPP_XOR1>(mn); mn = min_dpp<DPP_XOR2>(mn); mn = min_dpp<DPP_HALF_MIRROR>(mn);
    mn &= 0xfe00; mn |= mn << 16;
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = pk_sub(v[r], mn);
  }
  return 63 - (k & 63);
}

// two traceback chains per lane: chain id = lane + 64 q -> (decoder, call) = (id / V4_BLK, id % V4_BLK) of one block
struct V4Trace { int z[2], wsh[2], rowc[2]; bool ok[2]; long long ob[2]; };
__device__ __forceinline__ void v4_hop(V4Trace &T, const unsigned char *tab)
{
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const unsigned t = tab[(T.wsh[q] & 0x7e00) | T.rowc[q] | T.z[q]];        // state = path_byte >> 2 (d_viterbi.c:717)
    T.z[q] = (int)(t >> 2);
    T.wsh[q] -= 512;
  }
}
__device__ __forceinline__ void v4_trace_out(const V4Trace &T, const unsigned char *tab, uint8_t *out, long long out_lo)
{
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const unsigned t = tab[(T.wsh[q] & 0x7e00) | T.rowc[q] | T.z[q]];
    const int w = T.wsh[q] >> 9;
    const int sstart = rotl6(v4_cell_of_z((int)(t >> 2)), 2 * (((w % 3) + 3) % 3));    // phase of window w = (8w) % 6
    if (T.ok[q]) out[T.ob[q] - out_lo] = (unsigned char)((sstart << 2) | (t & 3u));
  }
}

// window WB of a block (WB = 0..11; the block starts at a window index that is a multiple of 6, so the window's phase is
// (8 WB) % 6).  HOPS: hops 2 WB and 2 WB + 1 of the previous block's traceback chains ride along -- which of them exist
// is a compile-time fact (NTB = ntraceback), so the window stays one straight-line block and the scheduler can spread
// the dependent LDS reads over it.
__device__ __forceinline__ void v4_load_w(unsigned (&W)[8], const unsigned *wrow, int wb)
{
  const uint4 *wp = reinterpret_cast<const uint4 *>(wrow + wb * 8);
  const uint4 t0 = wp[0], t1 = wp[1];
  W[0] = t0.x; W[1] = t0.y; W[2] = t0.z; W[3] = t0.w; W[4] = t1.x; W[5] = t1.y; W[6] = t1.z; W[7] = t1.w;
}
template <int WB, bool HOPS, int NTB> __device__ __forceinline__ void v4_fwd_window(int (&v)[4], const V4Lane &L, const unsigned *wrow, unsigned char *tab,
                                                                                  unsigned char *bests, int jb, int dd, int g, V4Trace &T,
                                                                                  const unsigned (&W)[8], unsigned (&Wnext)[8])
{
  constexpr int V6 = WB % 6;
  if (WB + 1 < V4_BLK) v4_load_w(Wnext, wrow, WB + 1);             // the next window's step words travel during this window
  if (HOPS && 2 * WB < NTB - 1) v4_hop(T, tab);
  const int jr = (jb + WB) & (V4_RINGW - 1);
  constexpr int P0 = (8 * V6) % 6;
  int raw[4];
  v4_window<P0>(v, W, L, raw);
  if (HOPS && 2 * WB + 1 < NTB - 1) v4_hop(T, tab);
  // the eight path bytes of this lane's cells = two words of the table (storage index z = 8*lane + 2r + h)
  *reinterpret_cast<uint2 *>(tab + ((jr * V4_DEC + dd) * 64 + g * 8)) =
      make_uint2(__builtin_amdgcn_perm((unsigned)raw[1], (unsigned)raw[0], 0x06040200u), __builtin_amdgcn_perm((unsigned)raw[3], (unsigned)raw[2], 0x06040200u));
  const int s = v4_window_end<(P0 + 2) % 6, (V6 & 1) == 1>(v, L);
  bests[dd * V4_RINGW + jr] = (unsigned char)s;                    // all 8 lanes of the group write the same byte
}
template <bool HOPS, int NTB> __device__ __forceinline__ void v4_fwd_block(int (&v)[4], const V4Lane &L, const unsigned *wrow, unsigned char *tab,
                                                                          unsigned char *bests, int jb, int dd, int g, V4Trace &T)
{
  unsigned Wa[8], Wb[8];
  v4_load_w(Wa, wrow, 0);
  v4_fwd_window<0, HOPS, NTB>(v, L, wrow, tab, bests, jb, dd, g, T, Wa, Wb); v4_fwd_window<1, HOPS, NTB>(v, L, wrow, tab, bests, jb, dd, g, T, Wb, Wa);
  v4_fwd_window<2, HOPS, NTB>(v, L, wrow, tab, bests, jb, dd, g, T, Wa, Wb); v4_fwd_window<3, HOPS, NTB>(v, L, wrow, tab, bests, jb, dd, g, T, Wb, Wa);
  v4_fwd_window<4, HOPS, NTB>(v, L, wrow, tab, bests, jb, dd, g, T, Wa, Wb); v4_fwd_window<5, HOPS, NTB>(v, L, wrow, tab, bests, jb, dd, g, T, Wb, Wa);
  v4_fwd_window<6, HOPS, NTB>(v, L, wrow, tab, bests, jb, dd, g, T, Wa, Wb); v4_fwd_window<7, HOPS, NTB>(v, L, wrow, tab, bests, jb, dd, g, T, Wb, Wa);
  v4_fwd_window<8, HOPS, NTB>(v, L, wrow, tab, bests, jb, dd, g, T, Wa, Wb); v4_fwd_window<9, HOPS, NTB>(v, L, wrow, tab, bests, jb, dd, g, T, Wb, Wa);
  v4_fwd_window<10, HOPS, NTB>(v, L, wrow, tab, bests, jb, dd, g, T, Wa, Wb); v4_fwd_window<11, HOPS, NTB>(v, L, wrow, tab, bests, jb, dd, g, T, Wb, Wa);
}

template <int NTB> __global__ __launch_bounds__(64) void viterbi4_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const RxState *st,
                                                      long long steps_fixed, VitParams vp, long long in_base, long long out_lo)
{
  __shared__ __attribute__((aligned(16))) unsigned char tab[V4_RINGW * V4_DEC * 64];   // path bytes: [window][decoder][cell z]
  __shared__ __attribute__((aligned(16))) unsigned wbuf[V4_DEC * V4_BLK * 8];          // step words: [decoder][step in block]
  __shared__ unsigned char bests[V4_DEC * V4_RINGW];
  __shared__ unsigned cbits[V4_DEC * V4_CBW];
  __shared__ unsigned lut[16];
  const int lane = threadIdx.x & 63, dd = lane >> 3, g = lane & 7;

  const long long total_steps = st ? st->n_vit_steps : steps_fixed;
  const long long total_out = total_steps / 8 - vp.ntb;
  const int B = vp.chunk_bytes, m = vp.m;
  constexpr int ntb = NTB;                                         // == vp.ntb (the host picks the instantiation)
  const long long chunk0 = (long long)blockIdx.x * V4_DEC;
  if (out_lo + chunk0 * B >= total_out) return;                   // whole wavefront idle
  const long long b0 = out_lo + (chunk0 + dd) * B;                // this lane's decoder
  const bool dec_active = b0 < total_out;
  const long long w0 = b0 + 2 - V4_WARM;                           // absolute window of relative window 0
  const int J = ((V4_WARM + B + ntb - 1 + V4_BLK - 1) / V4_BLK) * V4_BLK;
  const long long n_in_bytes = (total_steps * 2 / vp.plen * vp.n + vp.m - 1) / vp.m;   // input bytes that exist
  const int nload = ((2 * V4_BLK * 8 + 2 * m - 2) / m + 3 + 15) / 16;                  // lanes whose 16 bytes a block can need (7, 4, 3)

  if (lane < 16) {
    const int k0 = (lane >> 2) & 1, k1 = (lane >> 3) & 1, t1 = (lane >> 1) & 1, t0 = lane & 1;
    const int u0 = k0 ? 1 - 2 * t1 : 0, u1 = k1 ? 1 - 2 * (k0 ? t0 : t1) : 0;
    const unsigned d0 = (unsigned)(2 * (u0 + u1)) & 0xff, d1 = (unsigned)(2 * (-u0 + u1)) & 0xff;
    const unsigned d2 = (unsigned)(2 * (u0 - u1)) & 0xff, d3 = (unsigned)(2 * (-u0 - u1)) & 0xff;
    lut[lane] = d0 | (d1 << 8) | (d2 << 16) | (d3 << 24);        // see k_viterbi3.hpp
  }
  V4Lane L; v4_init_lane(g, L);
  int v[4] = {L.org[0], L.org[1], 0x01000100 | L.org[2], 0x01000100 | L.org[3]};   // phase 0: VGPRs 2,3 hold the upper states

  int ph0 = 0, bo0 = 0, off = 0; uint4 q = make_uint4(0, 0, 0, 0);
  auto stage_load = [&](int jb) {
    const long long tb = 8 * (w0 - 1) + (long long)jb * 8 - 2;     // real step index of block step 0 (may be < 0)
    const unsigned long long pbit = 2ull * (unsigned long long)(tb > 0 ? tb : 0);
    const unsigned long long pq = __umul64hi(pbit, vp.magic_plen);
    ph0 = (int)(pbit - pq * (unsigned)vp.plen);
    const unsigned long long rb = pq * (unsigned)vp.n + ((vp.prefix_nib >> (4 * ph0)) & 15ull);
    const unsigned long long by = __umul64hi(rb, vp.magic_m);
    const long long byte0 = (long long)by; bo0 = (int)(rb - by * (unsigned)m);
    off = (int)(((unsigned long long)(uintptr_t)in + (unsigned long long)(byte0 - in_base)) & 3ull);
    const long long src = byte0 - off + g * 16;                    // 4-byte aligned address
    q = make_uint4(0, 0, 0, 0);
    if (dec_active && g < nload) {
      if (src >= in_base && src + 16 <= n_in_bytes) q = *reinterpret_cast<const uint4 *>(in + (src - in_base));
      else {
        unsigned w[4] = {0, 0, 0, 0};
        for (int i = 0; i < 16; i++) {
          const long long bb = src + i;
          const unsigned bv = (bb >= in_base && bb < n_in_bytes) ? in[bb - in_base] : 0;
          w[i >> 2] |= bv << (8 * (i & 3));
        }
        q = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
  };
  auto stage_words = [&](int jb) {
    {
      const unsigned mk = (1u << m) - 1;
      auto grp = [&](unsigned d) { return ((d & mk) << (3 * m)) | (((d >> 8) & mk) << (2 * m)) | (((d >> 16) & mk) << m) | ((d >> 24) & mk); };
      const unsigned g0 = grp(q.x), g1 = grp(q.y), g2 = grp(q.z), g3 = grp(q.w);
      unsigned *cb = cbits + dd * V4_CBW;
      if (g < nload) {
        if (m == 2) cb[g] = (g0 << 24) | (g1 << 16) | (g2 << 8) | g3;
        else if (m == 4) { cb[2 * g] = (g0 << 16) | g1; cb[2 * g + 1] = (g2 << 16) | g3; }
        else { cb[3 * g] = (g0 << 8) | (g1 >> 16); cb[3 * g + 1] = (g1 << 16) | (g2 >> 8); cb[3 * g + 2] = (g2 << 24) | g3; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    constexpr int SPL = V4_BLK * 8 / 8;                            // steps per lane
    const long long tb = 8 * (w0 - 1) + (long long)jb * 8 - 2;
    const long long tbr = tb > 0 ? tb : 0;
    const int ub0 = g * SPL;
    const long long t = tb + ub0;
    int lo = 0;
    if (t < 0) lo = (-t < SPL) ? (int)(-t) : SPL;
    const long long rem = total_steps - t;
    int hi = !dec_active ? 0 : rem <= 0 ? 0 : rem < SPL ? (int)rem : SPL;
    if (hi < lo) hi = lo;
    int x = ph0 + 2 * (int)((t + lo) - tbr);
    if (x < 0) x = 0;
    const int dq = (int)(((unsigned)x * vp.magic16_plen) >> 16);
    const int ph = x - dq * vp.plen;
    const int pos = off * m + bo0 + dq * vp.n + (int)((vp.prefix_nib >> (4 * ph)) & 15ull) - (int)((vp.prefix_nib >> (4 * ph0)) & 15ull);
    const unsigned kmask = ((unsigned)(vp.punct_rep >> ph) << (2 * lo)) & (((1u << (2 * hi)) - 1u) & ~((1u << (2 * lo)) - 1u));
    const unsigned *cb = cbits + dd * V4_CBW;
    const unsigned cw0 = cb[pos >> 5], cw1 = cb[(pos >> 5) + 1];
    unsigned win = (unsigned)(((((unsigned long long)cw0) << 32) | cw1) >> (32 - (pos & 31)));
#pragma unroll
    for (int i = 0; i < SPL; i++) {
      const unsigned k2 = (kmask >> (2 * i)) & 3u;
      const unsigned idx = (k2 << 2) | (win >> 30);
      win <<= (k2 - (k2 >> 1));
      wbuf[dd * (V4_BLK * 8) + ub0 + i] = lut[idx];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  };
  V4Trace T;
  auto trace_init = [&](int jp) {
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const int cid = lane + 64 * c, tdec = cid / V4_BLK, call = cid - tdec * V4_BLK;
      const long long tb0 = out_lo + (chunk0 + tdec) * B;
      const long long tb1 = (tb0 + B < total_out) ? tb0 + B : total_out;
      int jj = jp + call;
      T.ob[c] = tb0 + (jj - (V4_WARM + ntb - 1));
      T.ok[c] = cid < V4_DEC * V4_BLK && tb0 < total_out && jj >= V4_WARM + ntb - 1 && T.ob[c] < tb1;
      const int td = cid < V4_DEC * V4_BLK ? tdec : 0;
      if (!T.ok[c]) jj = jp;                                       // any window inside the ring: result unused
      const int sb = bests[td * V4_RINGW + (jj & (V4_RINGW - 1))];
      T.z[c] = v4_z_of_cell(((sb | (sb << 6)) >> ((8 * jj + 8) % 6)) & 63);   // cell = rotr6(state, phase after the window)
      T.wsh[c] = jj << 9; T.rowc[c] = td * 64;
    }
  };

  stage_load(0);
  for (int jb = 0; jb < J; jb += V4_BLK) {
    stage_words(jb);
    if (jb + V4_BLK < J) stage_load(jb + V4_BLK);
    const bool tr = jb > 0 && !(vp.dbg & 1);
    if (tr) trace_init(jb - V4_BLK);
    if (!(vp.dbg & 2)) {
      const unsigned *wrow = wbuf + dd * (V4_BLK * 8);
      if (tr) v4_fwd_block<true, NTB>(v, L, wrow, tab, bests, jb, dd, g, T);
      else v4_fwd_block<false, NTB>(v, L, wrow, tab, bests, jb, dd, g, T);
    }
    if (tr) {
      if (vp.dbg & 2) for (int h = 0; h < ntb - 1; h++) v4_hop(T, tab);
      v4_trace_out(T, tab, out, out_lo);
    }
  }
  if (!(vp.dbg & 1)) {   // the last block's calls
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    trace_init(J - V4_BLK);
    for (int h = 0; h < ntb - 1; h++) v4_hop(T, tab);
    v4_trace_out(T, tab, out, out_lo);
  }
}

}  // namespace dvbt
