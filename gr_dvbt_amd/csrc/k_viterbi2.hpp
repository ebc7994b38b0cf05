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
// cell carries a tracking byte through the window: the cell its survivor occupied at the window
// start (6 bits) and the raw decisions of the window's last two steps (2 bits).  That is the same
// information: the reference hops with `state = path_byte >> 2` (:717), i.e. the state 8 steps
// back, and the decoded byte is (state 8 steps back) << 2 | the last two appended bits.
// The 64 tracking bytes per decoder per window go to an LDS ring; traceback = ntraceback-1 table
// hops from the window's best state (first index of the maximum, :699-711) + one decode.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "k_backend.hpp"

namespace dvbt {

constexpr int V2_WARM = 72;        // warm-up windows before a chunk's first byte
constexpr int V2_BLK = 24;         // windows per forward block (multiple of 3: the phase cycle of 6 steps vs 8-step windows)
constexpr int V2_RINGW = 48;       // windows kept in the LDS ring (>= V2_BLK + max ntraceback - 1)
constexpr int V2_WAVES = 2;        // wavefronts per workgroup
constexpr int V2_INBYTES = 256;    // input bytes staged per decoder per block (192 steps need <= 192 + slack)

#define DPP_XOR1 0xB1              /* quad_perm [1,0,3,2] */
#define DPP_XOR2 0x4E              /* quad_perm [2,3,0,1] */
#define DPP_HALF_MIRROR 0x141      /* lane i <-> 7-i inside each half row: logical bit a2 */
#define DPP_ROR8 0x128             /* row_ror:8: lane i <-> i^8 */
#define DPP_MIRROR 0x140

template <int CTRL> __device__ __forceinline__ int dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }

__device__ __forceinline__ int rotl6(int c, int p) { return ((c << p) | (c >> (6 - p))) & 63; }

struct V2Lane {                     // per-lane constants of the layout
  unsigned sel[6];                  // v_perm selectors: byte r = label class of slot r's butterfly at phase P
  int abit[4];                      // logical lane bits a0..a3
  unsigned kc[3][4];                // best-state key constants at window ends (phase 0,2,4): (63-state)<<6 | z
};

__device__ inline void v2_init_lane(int pl, V2Lane &L)
{
  int q = pl & 7, a2 = (q >> 2) & 1;
  int a = ((q ^ (a2 ? 7 : 0)) & 3) | (a2 << 2) | (pl & 8);
  for (int k = 0; k < 4; k++) L.abit[k] = (a >> k) & 1;
  for (int P = 0; P < 6; P++) {
    unsigned s = 0;
    for (int r = 0; r < 4; r++) {
      int c = r * 16 + a, st = rotl6(c, P), i = st & 31;
      int c0 = ((i >> 2) ^ (i >> 1) ^ i) & 1;                 // parity(2i & 0x4f)
      int c1 = ((i >> 4) ^ (i >> 2) ^ (i >> 1)) & 1;          // parity(2i & 0x6d)
      s |= (unsigned)(c0 | (c1 << 1)) << (8 * r);
    }
    L.sel[P] = s;
  }
  for (int e = 0; e < 3; e++)
    for (int r = 0; r < 4; r++) { int c = r * 16 + a; L.kc[e][r] = ((unsigned)(63 - rotl6(c, 2 * e)) << 6) | (unsigned)(r * 16 + pl); }
}

// one trellis step at compile-time phase P.  m[]: metrics 2M+bias, trk[]: survivor tracking byte, W: packed deltas.
// APPEND: the window's last two steps also shift the raw decision into the tracking byte.
template <int P, bool APPEND> __device__ __forceinline__ void v2_step(int (&m)[4], int (&trk)[4], unsigned W, const V2Lane &L)
{
  const unsigned E = __builtin_amdgcn_perm(W, W, L.sel[P]);
  int d[4], X[4], Y[4], Yp[4], Tp[4];
#pragma unroll
  for (int r = 0; r < 4; r++) { d[r] = (int)(E << (24 - 8 * r)) >> 24; X[r] = m[r] + d[r]; Y[r] = m[r] - d[r]; }
  if (P == 0) { Yp[0] = Y[2]; Yp[2] = Y[0]; Yp[1] = Y[3]; Yp[3] = Y[1]; Tp[0] = trk[2]; Tp[2] = trk[0]; Tp[1] = trk[3]; Tp[3] = trk[1]; }
  else if (P == 1) { Yp[0] = Y[1]; Yp[1] = Y[0]; Yp[2] = Y[3]; Yp[3] = Y[2]; Tp[0] = trk[1]; Tp[1] = trk[0]; Tp[2] = trk[3]; Tp[3] = trk[2]; }
  else {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      Yp[r] = P == 2 ? dpp<DPP_ROR8>(Y[r]) : P == 3 ? dpp<DPP_HALF_MIRROR>(Y[r]) : P == 4 ? dpp<DPP_XOR2>(Y[r]) : dpp<DPP_XOR1>(Y[r]);
      Tp[r] = P == 2 ? dpp<DPP_ROR8>(trk[r]) : P == 3 ? dpp<DPP_HALF_MIRROR>(trk[r]) : P == 4 ? dpp<DPP_XOR2>(trk[r]) : dpp<DPP_XOR1>(trk[r]);
    }
  }
  constexpr int PN = (P + 1) % 6;                               // bias for the next phase
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const bool from_partner = X[r] < Yp[r];                     // sign(T), T = X - Yp (never 0 when it matters: bias)
    const int sel = from_partner ? Tp[r] : trk[r];
    trk[r] = APPEND ? ((sel << 1) | (int)from_partner) : sel;
    const int mx = max(X[r], Yp[r]);
    const int nb = PN == 0 ? (r >> 1) : PN == 1 ? (r & 1) : PN == 2 ? L.abit[3] : PN == 3 ? L.abit[2] : PN == 4 ? L.abit[1] : L.abit[0];
    m[r] = (mx & ~1) | nb;
  }
}

// 8 steps of one window; P0 = phase of its first step (0, 2 or 4)
template <int P0> __device__ __forceinline__ void v2_window(int (&m)[4], int (&trk)[4], const unsigned (&W)[8], const V2Lane &L)
{
  v2_step<(P0 + 0) % 6, false>(m, trk, W[0], L); v2_step<(P0 + 1) % 6, false>(m, trk, W[1], L);
  v2_step<(P0 + 2) % 6, false>(m, trk, W[2], L); v2_step<(P0 + 3) % 6, false>(m, trk, W[3], L);
  v2_step<(P0 + 4) % 6, false>(m, trk, W[4], L); v2_step<(P0 + 5) % 6, false>(m, trk, W[5], L);
  v2_step<(P0 + 6) % 6, true>(m, trk, W[6], L);  v2_step<(P0 + 7) % 6, true>(m, trk, W[7], L);
}

template <int CTRL> __device__ __forceinline__ int row_min_step(int v) { return min(v, dpp<CTRL>(v)); }
template <int CTRL> __device__ __forceinline__ unsigned row_max_step(unsigned v) { return max(v, (unsigned)dpp<CTRL>((int)v)); }

// end of a window: min-renormalise, best state.  PE = phase after the window (0,2,4) -> kc index PE/2.
// returns the winning cell's storage index z (slot*16 + physical lane) in every lane of the row
template <int PE> __device__ __forceinline__ int v2_window_end(int (&m)[4], const V2Lane &L)
{
  int mn = min(min(m[0], m[1]), min(m[2], m[3]));
  mn = row_min_step<DPP_XOR1>(mn); mn = row_min_step<DPP_XOR2>(mn); mn = row_min_step<DPP_HALF_MIRROR>(mn); mn = row_min_step<DPP_MIRROR>(mn);
  mn &= ~1;
  unsigned key = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) { m[r] -= mn; key = max(key, ((unsigned)(m[r] >> 1) << 12) | L.kc[PE / 2][r]); }
  key = row_max_step<DPP_XOR1>(key); key = row_max_step<DPP_XOR2>(key); key = row_max_step<DPP_HALF_MIRROR>(key); key = row_max_step<DPP_MIRROR>(key);
  return (int)(key & 63);
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
  __shared__ __attribute__((aligned(16))) unsigned s_w[V2_WAVES][4 * V2_BLK * 8];   // packed deltas: [decoder][step in block]
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
  int m[4] = {0, 0, 1, 1};                                         // phase 0: slots 2,3 hold upper states
  int trk[4];

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
        // bytes: class 0 (c0=0,c1=0): u0+u1 | class 1 (c0=1): -u0+u1 | class 2 (c1=1): u0-u1 | class 3: -u0-u1,
        // times 2 so that a step with one punctured symbol (odd agreement difference) keeps the LSB free for the bias
        const unsigned Wd = ((unsigned)(2 * (u[0] + u[1])) & 0xff) | (((unsigned)(2 * (-u[0] + u[1])) & 0xff) << 8) |
                            (((unsigned)(2 * (u[0] - u[1])) & 0xff) << 16) | (((unsigned)(2 * (-u[0] - u[1])) & 0xff) << 24);
        wbuf[dd * (V2_BLK * 8) + ub0 + i] = Wd;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    // ---- forward: 24 windows, three phase variants per iteration
    for (int wi = 0; wi < V2_BLK && !(vp.dbg & 2); wi += 3) {
#pragma unroll
      for (int v3 = 0; v3 < 3; v3++) {
        const int j = jb + wi + v3;                                // relative window
        const int jr = j % V2_RINGW;
        unsigned W[8];
        {
          const uint4 *wp = reinterpret_cast<const uint4 *>(wbuf + dd * (V2_BLK * 8) + (wi + v3) * 8);
          uint4 lo = wp[0], hi = wp[1];
          W[0] = lo.x; W[1] = lo.y; W[2] = lo.z; W[3] = lo.w; W[4] = hi.x; W[5] = hi.y; W[6] = hi.z; W[7] = hi.w;
        }
#pragma unroll
        for (int r = 0; r < 4; r++) trk[r] = r * 16 + pl;          // survivors start in their own cell
        int z;
        if (v3 == 0) { v2_window<0>(m, trk, W, L); z = v2_window_end<2>(m, L); }     // steps 8j..8j+7 with 8j%6==0 -> ends at phase 2
        else if (v3 == 1) { v2_window<2>(m, trk, W, L); z = v2_window_end<4>(m, L); }
        else { v2_window<4>(m, trk, W, L); z = v2_window_end<0>(m, L); }
        if (pl == 0) bestz[dd * V2_RINGW + jr] = (unsigned char)z;
#pragma unroll
        for (int r = 0; r < 4; r++) tab[(jr * 4 + dd) * 64 + r * 16 + pl] = (unsigned char)trk[r];
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
      // decode window w: byte = (state at its start) << 2 | the two bits appended by its last two steps
      const unsigned t = tab[((w % V2_RINGW) * 4 + dd) * 64 + z];
      const int P0 = (8 * w) % 6, P6 = (P0 + 6) % 6, P7 = (P0 + 7) % 6;
      const int origin = (int)(t >> 2), s7 = (int)((t >> 1) & 1u), s8 = (int)(t & 1u);
      const int z7 = z ^ (s8 ? v2_xmask(P7) : 0);                 // cell the survivor occupied before the last step
      const int b8 = s8 ^ ((v2_logical(z) >> (5 - P7)) & 1);      // appended bit = decision ^ "own cell is an upper state"
      const int b7 = s7 ^ ((v2_logical(z7) >> (5 - P6)) & 1);
      const int sstart = rotl6(v2_logical(origin), P0);
      out[ob - out_lo] = (unsigned char)((sstart << 2) | (b7 << 1) | b8);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
}

}  // namespace dvbt
