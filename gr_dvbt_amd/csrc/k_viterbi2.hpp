// k_viterbi2.hpp -- A7, second-generation kernel: K=7 Viterbi with the add-compare-select
// butterflies on DPP lane exchanges (no LDS crossbar, no ds_bpermute in the inner loop).
//
// Mapping.  One wavefront decodes FOUR independent chunks; a chunk owns one DPP row (16 lanes) and
// every lane holds 4 of the 64 path metrics ("slots"), so 16 x 4 = 64 cells per decoder.
// The trellis is updated IN PLACE: the butterfly (i, i+32) -> (2i, 2i+1) is computed by the two
// cells that hold states i and i+32, each keeping one output, so after a step the cell that held
// state s holds state rotl6(s): cell c holds state rotl6(c, u mod 6) at relative step u.
// The two cells of a butterfly differ in exactly one bit of the cell index, which walks
// 5,4,3,2,1,0,5,... with the step; with the cell index laid out as
//     c = slot(2 bits) : a3 a2 a1 a0,   physical lane-in-row = (a0 + 2*a1) ^ (7*a2) ^ (8*a3)
// the six exchanges are: two in-register (slot bits) and four single DPP controls
// (row_ror:8, row_half_mirror, quad_perm[2,3,0,1], quad_perm[1,0,3,2]).
//
// Arithmetic.  Metrics are kept as 2*M + bias (M in units of half an agreement, so that the
// per-step increments stay even also when one of the two symbols is punctured); bias (the LSB) is
// 1 exactly when the cell currently holds an "upper" state (i+32), which makes every compare
// strict and reproduces the reference's tie rule (decision0/1 = (int8)(m0-m1) > 0, ties pick the
// i+32 predecessor: d_viterbi.c:508-521) with a single subtract; v_and_or re-arms the bias for the
// next step.  Branch metrics enter as delta = 2*(agreements - disagreements) of the butterfly's
// label with the received pair: X = M + delta (own), Y = M - delta (offered to the partner),
// new = max(X, Y_partner); this is the reference's m0..m3 up to an offset common to all 64 states,
// which the per-window min-renormalisation (d_viterbi.c:728-732) removes anyway.
// sign(T = X - Y_partner) is the survivor decision in cell space (1: came from the partner cell).
//
// Survivors.  Like the reference, which carries one path byte per state through each 8-step window
// and keeps a ring of the last `ntraceback` path arrays (ppresult, d_viterbi.c:68-77,690-693), every
// cell carries survivor tracking through the window -- in the LOW 16 bits of the metric register
// (metric in the high 16): the cell the survivor occupied at the window start, and the cell it
// occupied after the window's 6th step.  Because the bias makes the two candidates of a compare
// never equal in the metric field, max(X, Y_partner) on the whole register selects metric AND
// tracking of the winner in one instruction (and the DPP exchange folds into it).
// At the window end these give the reference's path byte: the reference hops with
// `state = path_byte >> 2` (:717), i.e. the state 8 steps back, and the byte's low two bits are the
// inputs of steps 7 and 8 = the top two bits of the state after step 6.
// The 64 tracking bytes per decoder per window go to an LDS ring; traceback = ntraceback-1 table
// hops from the window's best state (first index of the maximum, :699-711) + one decode.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include "k_backend.hpp"

namespace dvbt {

constexpr int V2_WARM = 72;        // warm-up windows before a chunk's first byte
constexpr int V2_BLK = 24;         // windows per forward block (multiple of 3: the phase cycle of 6 steps vs 8-step windows)
constexpr int V2_RINGW = 48;       // windows kept in the LDS ring (>= V2_BLK + max ntraceback - 1)
constexpr int V2_WAVES = 1;        // wavefronts per workgroup
constexpr int V2_INBYTES = 256;    // input bytes staged per decoder per block (192 steps need <= 192 + slack)

#define DPP_XOR1 0xB1              /* quad_perm [1,0,3,2] */
#define DPP_XOR2 0x4E              /* quad_perm [2,3,0,1] */
#define DPP_HALF_MIRROR 0x141      /* lane i <-> 7-i inside each half row: logical bit a2 */
#define DPP_ROR8 0x128             /* row_ror:8: lane i <-> i^8 */
#define DPP_MIRROR 0x140

template <int CTRL> __device__ __forceinline__ int dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }

__device__ __forceinline__ int rotl6(int c, int p) { return ((c << p) | (c >> (6 - p))) & 63; }

struct V2Lane {                     // per-lane constants of the layout
  unsigned sel[6][4];               // v_perm selectors: delta<<16 (sign extended) of slot r's butterfly label at phase P
  int abit[4];                      // logical lane bits a0..a3, pre-shifted to the bias position (bit 16)
  unsigned kc[3][4];                // best-state key constants at window ends (phase 0,2,4): (63-state)<<6 | z
};

__device__ inline void v2_init_lane(int pl, V2Lane &L)
{
  int q = pl & 7, a2 = (q >> 2) & 1;
  int a = ((q ^ (a2 ? 7 : 0)) & 3) | (a2 << 2) | (pl & 8);
  for (int k = 0; k < 4; k++) L.abit[k] = ((a >> k) & 1) << 16;
  // the step's two delta words hold the four label classes at byte positions 1 and 3:
  //   W0 = [0, d(class0), 0, d(class1)],  W1 = [0, d(class2), 0, d(class3)]
  // v_perm_b32(W1, W0, sel) with sel = {0x0c, 0x0c, byte index, sign-extension code} yields d << 16
  for (int P = 0; P < 6; P++)
    for (int r = 0; r < 4; r++) {
      int c = r * 16 + a, st = rotl6(c, P), i = st & 31;
      int c0 = ((i >> 2) ^ (i >> 1) ^ i) & 1;                 // parity(2i & 0x4f)
      int c1 = ((i >> 4) ^ (i >> 2) ^ (i >> 1)) & 1;          // parity(2i & 0x6d)
      int cls = c0 | (c1 << 1);
      unsigned idx = 1 + 2 * cls, code = 8 + cls;              // in[] = {W0 bytes 0..3, W1 bytes 0..3}
      L.sel[P][r] = 0x0c | (0x0c << 8) | (idx << 16) | (code << 24);
    }
  for (int e = 0; e < 3; e++)
    for (int r = 0; r < 4; r++) { int c = r * 16 + a; L.kc[e][r] = ((unsigned)(63 - rotl6(c, 2 * e)) << 6) | (unsigned)(r * 16 + pl); }
}

// one trellis step at compile-time phase P.  v[]: (2M+bias) << 16 | tracking, W0/W1: the step's delta words
template <int P> __device__ __forceinline__ void v2_step(int (&v)[4], unsigned W0, unsigned W1, const V2Lane &L)
{
  int X[4], Y[4], Yp[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int D = (int)__builtin_amdgcn_perm(W1, W0, L.sel[P][r]);     // delta << 16
    X[r] = v[r] + D; Y[r] = v[r] - D;
  }
  if (P == 0) { Yp[0] = Y[2]; Yp[2] = Y[0]; Yp[1] = Y[3]; Yp[3] = Y[1]; }
  else if (P == 1) { Yp[0] = Y[1]; Yp[1] = Y[0]; Yp[2] = Y[3]; Yp[3] = Y[2]; }
  else {
#pragma unroll
    for (int r = 0; r < 4; r++)
      Yp[r] = P == 2 ? dpp<DPP_ROR8>(Y[r]) : P == 3 ? dpp<DPP_HALF_MIRROR>(Y[r]) : P == 4 ? dpp<DPP_XOR2>(Y[r]) : dpp<DPP_XOR1>(Y[r]);
  }
  constexpr int PN = (P + 1) % 6;                               // bias for the next phase
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int mx = max(X[r], Yp[r]);                            // metric fields never tie (bias): tracking rides along
    const int nb = PN == 0 ? ((r >> 1) << 16) : PN == 1 ? ((r & 1) << 16) : PN == 2 ? L.abit[3] : PN == 3 ? L.abit[2] : PN == 4 ? L.abit[1] : L.abit[0];
    v[r] = (mx & ~0x10000) | nb;
  }
}

// 8 steps of one window; P0 = phase of its first step (0, 2 or 4).  Tracking: byte 1 = cell at the window
// start, byte 0 = cell after the 6th step.
template <int P0> __device__ __forceinline__ void v2_window(int (&v)[4], const unsigned (&W)[16], const V2Lane &L, int pl)
{
#pragma unroll
  for (int r = 0; r < 4; r++) v[r] = (v[r] & 0xffff0000) | ((r * 16 + pl) << 8);
  v2_step<(P0 + 0) % 6>(v, W[0], W[1], L); v2_step<(P0 + 1) % 6>(v, W[2], W[3], L);
  v2_step<(P0 + 2) % 6>(v, W[4], W[5], L); v2_step<(P0 + 3) % 6>(v, W[6], W[7], L);
  v2_step<(P0 + 4) % 6>(v, W[8], W[9], L); v2_step<(P0 + 5) % 6>(v, W[10], W[11], L);
#pragma unroll
  for (int r = 0; r < 4; r++) v[r] = (v[r] & 0xffffff00) | (r * 16 + pl);
  v2_step<(P0 + 6) % 6>(v, W[12], W[13], L); v2_step<(P0 + 7) % 6>(v, W[14], W[15], L);
}

template <int CTRL> __device__ __forceinline__ int row_min_step(int v) { return min(v, dpp<CTRL>(v)); }
template <int CTRL> __device__ __forceinline__ unsigned row_max_step(unsigned v) { return max(v, (unsigned)dpp<CTRL>((int)v)); }

// end of a window: best state (signed keys: the metric field may be negative between renormalisations) and, every
// fourth window, min-renormalisation (the reference subtracts the minimum at every output, d_viterbi.c:728-732; that
// only keeps its 8-bit metrics from wrapping -- decisions depend on differences -- and the 16-bit field here has room
// for 4 windows of drift).  PE = phase after the window (0,2,4) -> kc index PE/2.
// returns the winning cell's storage index z (slot*16 + physical lane) in every lane of the row
template <int PE> __device__ __forceinline__ int v2_window_end(int (&v)[4], const V2Lane &L, bool renorm)
{
  int key = INT_MIN;
#pragma unroll
  for (int r = 0; r < 4; r++) key = max(key, (int)(((unsigned)(v[r] >> 17) << 12) | L.kc[PE / 2][r]));
  key = max(key, dpp<DPP_XOR1>(key)); key = max(key, dpp<DPP_XOR2>(key)); key = max(key, dpp<DPP_HALF_MIRROR>(key)); key = max(key, dpp<DPP_MIRROR>(key));
  if (renorm) {
    int mn = min(min(v[0], v[1]), min(v[2], v[3]));             // ordering is decided by the metric field
    mn = row_min_step<DPP_XOR1>(mn); mn = row_min_step<DPP_XOR2>(mn); mn = row_min_step<DPP_HALF_MIRROR>(mn); mn = row_min_step<DPP_MIRROR>(mn);
    mn &= 0xfffe0000;                                           // metric without bias, tracking cleared
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] -= mn;
  }
  return key & 63;
}

// the reference's path byte from the tracking field: (cell at window start) << 2 | top two bits of the state after
// step 6.  After 6 steps of a window that began at phase P0 the phase is P0 again, so that state is
// rotl6(logical(z6), P0): its bits 5,4 are cell-index bits (5-P0, 4-P0).
template <int P0> __device__ __forceinline__ unsigned v2_track_byte(int v)
{
  const unsigned z6 = (unsigned)v & 0x3f, org = ((unsigned)v >> 8) & 0x3f;
  unsigned b;
  if (P0 == 0) b = z6 >> 4;                                     // slot bits
  else if (P0 == 2) b = (z6 >> 2) & 3;                          // a3 a2 = physical bits 3,2
  else b = (z6 & 3) ^ (((z6 >> 2) & 1) * 3);                    // a1 a0 = physical bits 1,0 ^ a2
  return (org << 2) | b;
}

// logical cell index (slot : a3..a0) of a storage index z (slot : physical lane)
__device__ __forceinline__ int v2_logical(int z)
{
  const int q = z & 7, a2 = (q >> 2) & 1;
  return (z & 0x38) | (a2 << 2) | ((q ^ (a2 ? 7 : 0)) & 3);
}
__device__ __forceinline__ int v2_xmask(int P) { return P == 0 ? 32 : P == 1 ? 16 : P == 2 ? 8 : P == 3 ? 7 : P == 4 ? 2 : 1; }

__global__ __launch_bounds__(64 * V2_WAVES) void viterbi2_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const RxState *st,
                                                                long long steps_fixed, VitParams vp, long long in_base, long long out_lo)
{
  __shared__ unsigned char s_tab[V2_WAVES][V2_RINGW * 4 * 64];     // tracking bytes: [window][decoder][cell z]  (the ppresult ring)
  __shared__ __attribute__((aligned(16))) unsigned s_w[V2_WAVES][4 * V2_BLK * 16];  // delta words W0,W1: [decoder][step in block][2]
  __shared__ __attribute__((aligned(16))) unsigned char s_in[V2_WAVES][4 * V2_INBYTES];   // staged input bytes per decoder
  __shared__ unsigned char s_best[V2_WAVES][4 * V2_RINGW];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, dd = lane >> 4, pl = lane & 15;
  unsigned char *tab = s_tab[wv]; unsigned *wbuf = s_w[wv]; unsigned char *inb = s_in[wv]; unsigned char *bestz = s_best[wv];

  const long long total_steps = st ? st->n_vit_steps : steps_fixed;
  const long long total_out = total_steps / 8 - vp.ntb;
  const int B = vp.chunk_bytes, ntb = vp.ntb;
  const long long chunk0 = ((long long)blockIdx.x * V2_WAVES + wv) * 4;
  if (out_lo + chunk0 * B >= total_out) return;                   // whole wavefront idle
  const long long b0 = out_lo + (chunk0 + dd) * B;                // this lane's decoder
  const bool dec_active = b0 < total_out;
  const long long b1 = (b0 + B < total_out) ? b0 + B : total_out;
  const long long w0 = b0 + 2 - V2_WARM;                           // absolute window of relative window 0
  const int J = ((V2_WARM + B + ntb - 1 + V2_BLK - 1) / V2_BLK) * V2_BLK;
  const long long n_in_bytes = (total_steps * 2 / vp.plen * vp.n + vp.m - 1) / vp.m;   // input bytes that exist

  V2Lane L; v2_init_lane(pl, L);
  int v[4] = {0, 0, 1 << 16, 1 << 16};                             // phase 0: slots 2,3 hold upper states

  for (int jb = 0; jb < J; jb += V2_BLK) {
    // ---- depuncture (viterbi_decoder_impl.cc:241-256) + delta packing for 192 steps x 4 decoders
    if (!(vp.dbg & 4)) {
      // position of a real step t: depunctured bit 2t -> (period q, phase ph) -> received-bit index rb
      // -> (input byte, bit offset).  Divisions by the small run-time constants 2k and m use
      // host-made magic multipliers (exact far beyond any segment length).
      auto locate = [&](long long t, int &ph, long long &byte, int &bo) {
        const unsigned long long pbit = 2ull * (unsigned long long)t;
        const unsigned long long q = __umul64hi(pbit, vp.magic_plen);
        ph = (int)(pbit - q * (unsigned)vp.plen);
        const unsigned long long rb = q * (unsigned)vp.n + ((vp.prefix_nib >> (4 * ph)) & 15ull);
        const unsigned long long by = __umul64hi(rb, vp.magic_m);
        byte = (long long)by; bo = (int)(rb - by * (unsigned)vp.m);
      };
      const long long tb = 8 * (w0 - 1) + (long long)jb * 8 - 2;   // real step index of block step 0 (may be < 0)
      int ph0, bo0; long long byte0;
      locate(tb > 0 ? tb : 0, ph0, byte0, bo0);                    // same value in the 16 lanes of the row
      {   // cooperative load of V2_INBYTES input bytes per decoder (16 per lane)
        const long long src = byte0 + pl * 16;
        unsigned char tmp[16];
#pragma unroll
        for (int i = 0; i < 16; i++) { long long bb = src + i; tmp[i] = (dec_active && bb < n_in_bytes) ? in[bb - in_base] : 0; }
#pragma unroll
        for (int i = 0; i < 16; i++) inb[dd * V2_INBYTES + pl * 16 + i] = tmp[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      // 12 steps per lane, branch-free incremental depuncture
      const int ub0 = pl * 12;
      long long t = tb + ub0;
      int ph, bo; long long byte;
      locate(t > 0 ? t : 0, ph, byte, bo);
      int boff = (int)(byte - byte0);
      const unsigned char *ib = inb + dd * V2_INBYTES;
#pragma unroll
      for (int i = 0; i < 12; i++, t++) {
        const bool real = dec_active && t >= 0 && t < total_steps;
        int u[2];
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
          const int kept = real ? (int)((vp.punct_mask >> ph) & 1u) : 0;
          const int bit = (ib[boff & (V2_INBYTES - 1)] >> (vp.m - 1 - bo)) & 1;
          u[hh] = kept * (1 - 2 * bit);
          bo += kept;
          const int wrap = bo == vp.m;
          bo = wrap ? 0 : bo; boff += wrap;
          if (real) { ph++; ph = ph == vp.plen ? 0 : ph; }
        }
        // class 0 (c0=0,c1=0): u0+u1 | class 1 (c0=1): -u0+u1 | class 2 (c1=1): u0-u1 | class 3: -u0-u1, times 2 so
        // that a step with one punctured symbol (odd agreement difference) keeps the LSB free for the bias;
        // stored at byte positions 1 and 3 (the ones v_perm can sign-extend)
        const unsigned Wa = (((unsigned)(2 * (u[0] + u[1])) & 0xff) << 8) | (((unsigned)(2 * (-u[0] + u[1])) & 0xff) << 24);
        const unsigned Wb = (((unsigned)(2 * (u[0] - u[1])) & 0xff) << 8) | (((unsigned)(2 * (-u[0] - u[1])) & 0xff) << 24);
        *reinterpret_cast<uint2 *>(wbuf + (dd * (V2_BLK * 8) + ub0 + i) * 2) = make_uint2(Wa, Wb);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    // ---- forward: 24 windows, three phase variants per iteration
    for (int wi = 0; wi < V2_BLK && !(vp.dbg & 2); wi += 3) {
#pragma unroll
      for (int v3 = 0; v3 < 3; v3++) {
        const int j = jb + wi + v3;                                // relative window
        const int jr = j % V2_RINGW;
        unsigned W[16];
        {
          const uint4 *wp = reinterpret_cast<const uint4 *>(wbuf + (dd * (V2_BLK * 8) + (wi + v3) * 8) * 2);
#pragma unroll
          for (int q4 = 0; q4 < 4; q4++) { uint4 t4 = wp[q4]; W[4 * q4] = t4.x; W[4 * q4 + 1] = t4.y; W[4 * q4 + 2] = t4.z; W[4 * q4 + 3] = t4.w; }
        }
        int z; unsigned tb[4];
        if (v3 == 0) {            // steps 8j..8j+7 with 8j%6==0 -> ends at phase 2
          v2_window<0>(v, W, L, pl);
#pragma unroll
          for (int r = 0; r < 4; r++) tb[r] = v2_track_byte<0>(v[r]);
          z = v2_window_end<2>(v, L, (j & 3) == 3);
        } else if (v3 == 1) {
          v2_window<2>(v, W, L, pl);
#pragma unroll
          for (int r = 0; r < 4; r++) tb[r] = v2_track_byte<2>(v[r]);
          z = v2_window_end<4>(v, L, (j & 3) == 3);
        } else {
          v2_window<4>(v, W, L, pl);
#pragma unroll
          for (int r = 0; r < 4; r++) tb[r] = v2_track_byte<4>(v[r]);
          z = v2_window_end<0>(v, L, (j & 3) == 3);
        }
        if (pl == 0) bestz[dd * V2_RINGW + jr] = (unsigned char)z;
#pragma unroll
        for (int r = 0; r < 4; r++) tab[(jr * 4 + dd) * 64 + r * 16 + pl] = (unsigned char)tb[r];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    // ---- traceback for the 24 calls x 4 decoders of this block (d_viterbi.c:714-724), lane = (decoder, call)
    for (int pass = 0; pass < 2 && !(vp.dbg & 1); pass++) {
      const int jj = pass == 0 ? pl : 16 + pl;
      if (jj >= V2_BLK) continue;
      const int j = jb + jj;
      const long long ob = b0 + (j - (V2_WARM + ntb - 1));         // output byte of relative call j
      if (!dec_active || j < V2_WARM + ntb - 1 || ob >= b1) continue;
      int z = bestz[dd * V2_RINGW + (j % V2_RINGW)];
      int w = j;
      for (int hop = 0; hop < ntb - 1; hop++, w--) z = tab[((w % V2_RINGW) * 4 + dd) * 64 + z] >> 2;   // state = path_byte >> 2 (:717)
      // decode window w: byte = (state at its start) << 2 | inputs of its steps 7 and 8 (already in the table byte)
      const unsigned t = tab[((w % V2_RINGW) * 4 + dd) * 64 + z];
      const int sstart = rotl6(v2_logical((int)(t >> 2)), (8 * w) % 6);
      out[ob - out_lo] = (unsigned char)((sstart << 2) | (t & 3u));
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
}

}  // namespace dvbt
