// k_soft4.hpp -- the soft-input decoder on the hard kernel's cell layout (k_viterbi3.hpp): one wavefront decodes FOUR chunks, a chunk owns one DPP row,
// every lane holds 4 of the 64 path metrics as the 16-bit halves of two VGPRs, the trellis is updated in place (cell c holds state rotl6(c, u mod 6) at
// relative step u; the six butterfly exchanges are a VGPR swap, a half swap and four DPP controls).  What differs from the hard kernel:
//   * a cell is a full 16-bit metric (soft values in [-31, 31]: a branch delta is up to +-62, the spread of the 64 metrics up to ~750; the row maximum is
//     subtracted every 48 steps), so the survivors cannot ride along in a path byte: the DECISION of every cell and step (1 = the survivor came from the
//     butterfly partner) is the sign of X - Y_partner, moved to the step's bit of a register per VGPR (v_pk_lshrrev_b16 + v_bfi) and, every 8 steps, packed to one dword per
//     lane (4 cells x 8 steps) and written to the wavefront's slot of a scratch buffer in HBM (256 coalesced bytes per 8 steps: written once, read once);
//   * the branch deltas of a step are the four correlations +-sx +-sy: one word (A = sx + sy, B = sy - sx) per step and decoder from LDS, its negation,
//     and the hard kernel's per-lane v_perm selectors pick every cell's class out of the 8 bytes;
//   * the traceback runs as 8 segments per decoder, a lane per segment, each from the best cell recorded at a block end 192 steps behind the segment: the
//     cell walks in PHYSICAL coordinates z = r : h : row : lane-in-row, where every butterfly partner is an XOR with a constant (0x80, 0x40, 8, 7, 2, 1 for
//     phases 0..5); the decision rows of the 8 segments pass through LDS.
// The soft values arrive de-interleaved from soft_demap_kernel; the staging of a block depunctures them (viterbi_decoder_impl.cc:241-256, erasures = 0).
// The first version of the soft decoder (one wavefront per chunk, a lane per state, two ds_bpermute per step; 5.0 ms on 17 superframes of 8k QAM64 7/8
// after its decisions had moved from LDS to HBM, 27.9 ms before) was bound by the LDS crossbar: 72 cycles per step and SIMD for 8 VALU instructions.
#pragma once
#include "k_soft.hpp"
#ifndef S4_EXP
#define S4_EXP 0     /* attribution builds (tools/): 1 = no traceback, 2 = no forward pass */
#endif

namespace dvbt {

constexpr int S4_WARM = 256;               // warm-up steps in front of a chunk
constexpr int S4_BLK = 48;                 // steps per staged block: a multiple of the 6-step phase cycle and of the 8-step decision group
constexpr int S4_WAVES = 4;                // wavefronts per workgroup
constexpr int S4_GRID = 2048;              // workgroups (8 wavefronts per SIMD)
constexpr int S4_BMAX = 304;               // largest chunk (decoded bytes)
constexpr int S4_LOOK = 128;               // steps decoded behind a chunk, at least (the reference's depth 8 ntraceback is 40 steps at rate 1/2: the output
                                           // DELAY stays the reference's, the decision depth need not)
constexpr int S4_G0 = 30;                  // first decision group that is kept (groups of the warm-up are never traced; a multiple of 6 below S4_WARM / 8)
#ifndef S4_NSEG_N
#define S4_NSEG_N 8                        /* tools/soft_nseg.py builds 4 beside it: the re-read factor of the decisions against the length of the dependent traceback chains */
#endif
constexpr int S4_NSEG = S4_NSEG_N;         // traceback segments per decoder
constexpr int S4_PRE = 24;                 // groups a segment's chain starts behind the segment's end (192 steps = the reference's depth at rate 7/8)
constexpr int S4_MAXSTEPS = ((S4_WARM + 8 * S4_BMAX + 8 * 24 + 16 + S4_BLK - 1) / S4_BLK) * S4_BLK;
static_assert(((S4_WARM + 8 * S4_BMAX + 8 * 24 + S4_BLK - 1) / S4_BLK) * S4_BLK <= S4_MAXSTEPS, "the largest plan (s4_plan: B = S4_BMAX, look = 8 x 24) must fit the decision slots and the best-cell table");
constexpr size_t S4_SLOT_WORDS = (size_t)(S4_MAXSTEPS / 8 - S4_G0) * 64;    // dwords of decisions per wavefront
constexpr size_t S4_SCRATCH_WORDS = (size_t)S4_GRID * S4_WAVES * S4_SLOT_WORDS;

// host: chunk size and steps per decoder for a stream of total_out bytes: whole rounds of the wavefront slots; the steps are rounded up to whole blocks
// (the look-ahead grows by up to 40 steps)
struct S4Plan { int B, nsteps; };
inline S4Plan s4_plan(long long total_out, int ntb)
{
  const long long per_round = 4ll * S4_GRID * S4_WAVES;             // decoders resident
  long long rounds = (total_out + per_round * 256 / 2) / (per_round * 256); if (rounds < 1) rounds = 1;
  long long B;
  for (;;) {
    B = (total_out + rounds * per_round - 1) / (rounds * per_round);
    if (B < 64) B = 64;
    if (B <= S4_BMAX) break;
    rounds++;
  }
  const int look = 8 * ntb > S4_LOOK ? 8 * ntb : S4_LOOK;
  S4Plan p; p.B = (int)B; p.nsteps = ((S4_WARM + 8 * p.B + look + S4_BLK - 1) / S4_BLK) * S4_BLK;
  return p;
}

// workgroups of the launch for a stream of total_out bytes.  NOT monotone in total_out: s4_plan raises B with the stream's length, so the task count drops
// each time B increments (total = 64 * 32768: 2048 workgroups, one byte more: 2017).
inline unsigned s4_grid(long long total_out, int ntb)
{
  const S4Plan sp = s4_plan(total_out, ntb);
  const long long tasks = (total_out + 4ll * sp.B - 1) / (4ll * sp.B);
  const long long g = (tasks + S4_WAVES - 1) / S4_WAVES;
  return (unsigned)(g < 1 ? 1 : (g > S4_GRID ? S4_GRID : g));
}
// the largest grid any stream of at most max_total bytes launches (B >= 64 always): what a handle's decision scratch is sized for.  The launch is clamped to
// it as well (the kernel strides its tasks by gridDim), so the scratch can never be indexed beyond its end whatever s4_plan does
inline unsigned s4_grid_bound(long long max_total)
{
  const long long tasks = (max_total + 255) / 256, g = (tasks + S4_WAVES - 1) / S4_WAVES;
  return (unsigned)(g < 1 ? 1 : (g > S4_GRID ? S4_GRID : g));
}

struct S4Lane { unsigned sel[6][2]; };     // v_perm selectors: the 16-bit class delta of the lo / hi cell of VGPR r at phase P out of [A, B, -A, -B]
__device__ inline void s4_init_lane(int pl, S4Lane &L)
{
  const int a = v3_log(pl);
  for (int r = 0; r < 2; r++)
    for (int P = 0; P < 6; P++) {
      unsigned b[2];
      for (int h = 0; h < 2; h++) {
        const int c = (r << 5) | (h << 4) | a, i = rotl6(c, P) & 31;
        const int c0 = ((i >> 2) ^ (i >> 1) ^ i) & 1;                 // parity(2i & 0x4f)
        const int c1 = ((i >> 4) ^ (i >> 2) ^ (i >> 1)) & 1;          // parity(2i & 0x6d)
        // class 0: sx + sy = A (bytes 0,1) | class 1 (c0 = 1): -sx + sy = B (2,3) | class 2 (c1 = 1): sx - sy = -B (6,7) | class 3: -A (4,5)
        const int cls = c0 | (c1 << 1);
        b[h] = cls == 0 ? 0u : cls == 1 ? 2u : cls == 2 ? 6u : 4u;
      }
      L.sel[P][r] = b[0] | ((b[0] + 1) << 8) | (b[1] << 16) | ((b[1] + 1) << 24);
    }
}

// (mask & a) | (~mask & b): one v_bfi_b32 (left to itself the compiler emits v_and + v_and_or)
__device__ __forceinline__ int s4_bfi(unsigned mask, unsigned a, unsigned b) { int r; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(mask), "v"(a), "v"(b)); return r; }

typedef unsigned short s4u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned s4_pk_lshr(unsigned x, int n) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(s4u2, x) >> (unsigned short)n); }   // v_pk_lshrrev_b16

// one step at phase P: W = A | B << 16 of this lane's decoder, nW = -A | -B << 16
template <int P, int JJ> __device__ __forceinline__ void s4_step(int (&v)[2], int (&dacc)[2], unsigned W, unsigned nW, const S4Lane &L)   // JJ: step within the group of 8
{
  int X[2], Y[2], Yp[2];
  const int D0 = (int)__builtin_amdgcn_perm(nW, W, L.sel[P][0]);
  X[0] = pk_add(v[0], D0); Y[0] = pk_sub(v[0], D0);
  // VGPR index = cell bit 5 = state bit (5 + P) % 6: at phases 0 and 4 both VGPRs have the same deltas, at phases 2 and 3 VGPR 1 the negated ones (k_viterbi3.hpp)
  if (P == 0 || P == 4) { X[1] = pk_add(v[1], D0); Y[1] = pk_sub(v[1], D0); }
  else if (P == 2 || P == 3) { X[1] = pk_sub(v[1], D0); Y[1] = pk_add(v[1], D0); }
  else {
    const int D1 = (int)__builtin_amdgcn_perm(nW, W, L.sel[P][1]);
    X[1] = pk_add(v[1], D1); Y[1] = pk_sub(v[1], D1);
  }
  if (P == 0) { Yp[0] = Y[1]; Yp[1] = Y[0]; }
  else {
#pragma unroll
    for (int r = 0; r < 2; r++)
      Yp[r] = P == 1 ? (int)__builtin_amdgcn_alignbit((unsigned)Y[r], (unsigned)Y[r], 16) : P == 2 ? dppb<DPP_ROR8>(Y[r]) : P == 3 ? dppb<DPP_HALF_MIRROR>(Y[r])
                     : P == 4 ? dppb<DPP_XOR2>(Y[r]) : dppb<DPP_XOR1>(Y[r]);
  }
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int diff = pk_sub(X[r], Yp[r]);                            // sign set: the partner's offer wins
    v[r] = pk_max(X[r], Yp[r]);
    // the sign of both halves to bit 8 + JJ of its half (bits 8..15 of a half = the group's 8 decisions, oldest lowest): the halves' shift, then v_bfi
    // replaces exactly that bit (whatever the previous group left there); the last step of a group needs no shift
    const unsigned sh = JJ == 7 ? (unsigned)diff : s4_pk_lshr((unsigned)diff, 7 - JJ);
    dacc[r] = s4_bfi(0x01000100u << JJ, sh, (unsigned)dacc[r]);
  }
}
template <int P0> __device__ __forceinline__ void s4_group(int (&v)[2], int (&dacc)[2], const unsigned *w, const S4Lane &L)   // w: W, nW of 8 steps
{
  s4_step<(P0 + 0) % 6, 0>(v, dacc, w[0], w[1], L); s4_step<(P0 + 1) % 6, 1>(v, dacc, w[2], w[3], L); s4_step<(P0 + 2) % 6, 2>(v, dacc, w[4], w[5], L);
  s4_step<(P0 + 3) % 6, 3>(v, dacc, w[6], w[7], L); s4_step<(P0 + 4) % 6, 4>(v, dacc, w[8], w[9], L); s4_step<(P0 + 5) % 6, 5>(v, dacc, w[10], w[11], L);
  s4_step<(P0 + 6) % 6, 6>(v, dacc, w[12], w[13], L); s4_step<(P0 + 7) % 6, 7>(v, dacc, w[14], w[15], L);
}

// traceback of the 8 steps of one decision group (GI = group index mod 3: the phases are static): z = physical cell (lane | h << 6 | r << 7) after the
// group's last step, updated to the cell before its first; returns the decoded byte (MSB = first step).  row = the group's decision dwords in LDS; the dword of
// the cell's lane is read again only when the lane can have changed (the previous flip was a DPP phase) or the group has.
// (Earlier versions: four scalar chains with v_readlane -- ~7 SALU per step and chain, and the scalar unit issues one instruction per four cycles and SIMD
// like the vector unit: 2.65 ms of traceback against 1.66 ms of forward pass on 17 superframes; then lane k < 4 = the whole chain of decoder k with
// ds_bpermute: 0.75 ms, still 2240 dependent steps x 9 instructions for 4 useful lanes.)
template <int GI, int JJ> __device__ __forceinline__ void s4_back(int &z, unsigned &byte, const unsigned *row, unsigned &word)
{
  constexpr int J = GI * 8 + JJ, P = J % 6, P1 = (J + 1) % 6;
  // decoded bit of step J = LSB of the state after it = cell bit (6 - P1) % 6 of the cell after it: a0, r, h, a3, a2, a1 for P1 = 0..5
  const unsigned zz = (unsigned)z;
  const unsigned bit = P1 == 0 ? (zz ^ (zz >> 2)) & 1u : P1 == 1 ? (zz >> 7) & 1u : P1 == 2 ? (zz >> 6) & 1u : P1 == 3 ? (zz >> 3) & 1u : P1 == 4 ? (zz >> 2) & 1u
                                                                                                                                       : ((zz >> 1) ^ (zz >> 2)) & 1u;
  byte |= bit << (7 - JJ);
  if (JJ == 7 || P1 >= 2) word = row[z & 63];
  const unsigned d = (word >> ((zz >> 6) * 8 + JJ)) & 1u;
  constexpr unsigned mask = P == 0 ? 0x80 : P == 1 ? 0x40 : P == 2 ? 8 : P == 3 ? 7 : P == 4 ? 2 : 1;
  z ^= (int)(d * mask);
  if constexpr (JJ > 0) s4_back<GI, JJ - 1>(z, byte, row, word);
}

__global__ __launch_bounds__(64 * S4_WAVES) void viterbi_soft4_kernel(const int8_t *__restrict__ soft, uint8_t *__restrict__ out, const RxState *st, VitParams vp,
                                                                      unsigned *__restrict__ scratch, int B, int nsteps)
{
  __shared__ __attribute__((aligned(16))) unsigned wbuf_[S4_WAVES][2][8 * S4_BLK];               // step words W, -W [buffer][decoder][step in block]
  __shared__ unsigned rows_[S4_WAVES][S4_NSEG * 64];                                             // traceback: one decision row per segment
  __shared__ unsigned char best_[S4_WAVES][4 * (S4_MAXSTEPS / S4_BLK)];                          // best cell of every decoder after every block
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, dd = lane >> 4, pl = lane & 15;
  if (st->first_out < 0) return;
  unsigned *rows = rows_[wv]; unsigned char *bests = best_[wv];
  const long long total_steps = st->n_vit_steps, total_out = total_steps / 8 - vp.ntb, n_soft = st->n_vit_in * vp.m;
  const unsigned magic32 = 0xffffffffu / (unsigned)vp.plen + 1u;   // x / plen == umulhi(x, magic32) for x * plen < 2^32
  unsigned *dec = scratch + ((size_t)blockIdx.x * S4_WAVES + wv) * S4_SLOT_WORDS;
  unsigned (*wb)[8 * S4_BLK] = wbuf_[wv];
  S4Lane L; s4_init_lane(pl, L);
  const int nblk = nsteps / S4_BLK, ngrp = nsteps / 8;
  const long long nslots = (long long)gridDim.x * S4_WAVES;
  for (long long task = (long long)blockIdx.x * S4_WAVES + wv; task * 4 * B < total_out; task += nslots) {
    const long long b0 = (task * 4 + dd) * B;                      // this lane's decoder
    const long long t0 = 8 * b0 - S4_WARM;                          // its first step (may be < 0: erasures)
    const bool active = b0 < total_out;
    // depuncturing relative to the decoder's first real step: one 64-bit locate per task, then 32-bit arithmetic (x < 2^13: x / plen by one multiply)
    const long long tbase = t0 > 0 ? t0 : 0;
    const int skip = (int)(tbase - t0);                             // leading steps before the stream's start: erasures
    const unsigned long long pbit = 2ull * (unsigned long long)tbase, pq = __umul64hi(pbit, vp.magic_plen);
    const int phb = (int)(pbit - pq * (unsigned)vp.plen);
    const long long rb = (long long)(pq * (unsigned)vp.n);
    auto load3 = [&](int blk, int (&q)[3]) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int ir = blk * S4_BLK + pl + 16 * k - skip;
        int sx = 0, sy = 0;
        if (active && ir >= 0 && tbase + ir < total_steps) {
          const unsigned x = (unsigned)(phb + 2 * ir), dq = __umulhi(x, magic32); const int ph = (int)(x - dq * vp.plen);
          const long long r = rb + (long long)(dq * (unsigned)vp.n);
          if ((vp.punct_mask >> ph) & 1) { const long long rr = r + (int)((vp.prefix_nib >> (4 * ph)) & 15); if (rr < n_soft) sx = soft[rr]; }
          if ((vp.punct_mask >> (ph + 1)) & 1) { const long long rr = r + (int)((vp.prefix_nib >> (4 * (ph + 1))) & 15); if (rr < n_soft) sy = soft[rr]; }
        }
        q[k] = (sx & 0xff) | (sy << 8);
      }
    };
    auto put3 = [&](int buf, const int (&q)[3]) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int sx = (int)(signed char)(q[k] & 0xff), sy = q[k] >> 8;
        const unsigned W = (unsigned)((sx + sy) & 0xffff) | ((unsigned)(sy - sx) << 16), nW = (unsigned)((-sx - sy) & 0xffff) | ((unsigned)(sx - sy) << 16);
        *reinterpret_cast<uint2 *>(&wb[buf][2 * (dd * S4_BLK + pl + 16 * k)]) = make_uint2(W, nW);
      }
    };
    int v[2] = {0, 0}, dacc[2] = {0, 0};
    int q[3];
    load3(0, q); put3(0, q);
    for (int blk = 0; blk < ((S4_EXP & 2) ? 1 : nblk); blk++) {
      if (blk + 1 < nblk) load3(blk + 1, q);
      const unsigned *wrow = &wb[blk & 1][2 * dd * S4_BLK];
#pragma unroll
      for (int gi = 0; gi < 6; gi++) {
        unsigned w[16];
        { const uint4 *wp = reinterpret_cast<const uint4 *>(wrow + gi * 16);
#pragma unroll
          for (int i4 = 0; i4 < 4; i4++) { const uint4 x = wp[i4]; w[4 * i4] = x.x; w[4 * i4 + 1] = x.y; w[4 * i4 + 2] = x.z; w[4 * i4 + 3] = x.w; } }
        if (gi % 3 == 0) s4_group<0>(v, dacc, w, L); else if (gi % 3 == 1) s4_group<2>(v, dacc, w, L); else s4_group<4>(v, dacc, w, L);
        const int g = blk * 6 + gi;
        // the four cells' decisions of the group: bit jj of byte 2 r + h = step jj
        if (g >= S4_G0) dec[(size_t)(g - S4_G0) * 64 + lane] = __builtin_amdgcn_perm((unsigned)dacc[1], (unsigned)dacc[0], 0x07050301u);
      }
      {   // the row's best metric back to 0
        int k = pk_max(v[0], v[1]);
        k = pk_max(k, dppb<DPP_XOR1>(k)); k = pk_max(k, dppb<DPP_XOR2>(k)); k = pk_max(k, dppb<DPP_HALF_MIRROR>(k)); k = pk_max(k, dppb<DPP_MIRROR>(k));
        k = pk_max(k, (int)__builtin_amdgcn_alignbit((unsigned)k, (unsigned)k, 16));
        v[0] = pk_sub(v[0], k); v[1] = pk_sub(v[1], k);
        // a best cell of the row (metric 0 now), as a physical cell: the traceback segments start from it.  key = lane-in-row | h << 6 | r << 7, 0x100 = none
        const int mine = (short)(v[0] & 0xffff) == 0 ? 0 : (v[0] >> 16) == 0 ? 0x40 : (short)(v[1] & 0xffff) == 0 ? 0x80 : (v[1] >> 16) == 0 ? 0xc0 : 0x100;
        int key = mine | pl;
        key = min_dpp<DPP_XOR1>(key); key = min_dpp<DPP_XOR2>(key); key = min_dpp<DPP_HALF_MIRROR>(key); key = min_dpp<DPP_MIRROR>(key);
        if (pl == 0) bests[blk * 4 + dd] = (unsigned char)((key & 0xcf) | (dd << 4));
      }
      if (blk + 1 < nblk) put3((blk + 1) & 1, q);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    // traceback in S4_NSEG segments per decoder, lane = (segment, decoder) (lanes 32..63 duplicate 0..31 and store nothing): a segment of S groups starts
    // S4_PRE groups behind its end, from the best cell recorded there, so that 32 chains of S + S4_PRE groups replace 4 chains of the whole chunk.  Segment
    // bounds, S and S4_PRE are multiples of 6 groups (= block ends, and every lane is at the same phase).  Per group: the 8 segments' decision rows go
    // through LDS (loaded one group ahead), then 8 steps per lane.
    {
      const int k = lane & 3, sg = (lane >> 2) & (S4_NSEG - 1);
      const int S = ((S4_WARM / 8 + B - S4_G0 + S4_NSEG * 6 - 1) / (S4_NSEG * 6)) * 6;
      auto top_of = [&](int s_) { const int t_ = S4_G0 + (s_ + 1) * S + S4_PRE; return t_ < ngrp ? t_ : ngrp; };
      const int lo = S4_G0 + sg * S, hi = lo + S, top = top_of(sg);
      int z = bests[(top / 6 - 1) * 4 + k];
      const long long ob0 = (task * 4 + k) * B, ob1 = ob0 + B < total_out ? ob0 + B : total_out;
      const int nsub = (S4_EXP & 1) ? 3 : S + S4_PRE;               // groups per chain
      unsigned nxt[S4_NSEG];
      auto load_rows = [&](int sub) {                               // the row of every segment for its group number `sub` from the top
#pragma unroll
        for (int s_ = 0; s_ < S4_NSEG; s_++) {
          int g = top_of(s_) - 1 - sub; g = g < S4_G0 ? S4_G0 : g;
          nxt[s_] = __hip_atomic_load(&dec[(size_t)(g - S4_G0) * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      };
      load_rows(0);
      for (int sub = 0; sub < nsub; sub += 3) {
#pragma unroll
        for (int gi = 2; gi >= 0; gi--) {                           // top - 1 - sub is 2 (mod 3)
#pragma unroll
          for (int s_ = 0; s_ < S4_NSEG; s_++) rows[s_ * 64 + lane] = nxt[s_];
          const int su = sub + (2 - gi);
          if (su + 1 < nsub) load_rows(su + 1);
          const int g = top - 1 - su;
          unsigned byte = 0, word = 0;
          const unsigned *row = rows + sg * 64;
          if (gi == 2) s4_back<2, 7>(z, byte, row, word); else if (gi == 1) s4_back<1, 7>(z, byte, row, word); else s4_back<0, 7>(z, byte, row, word);
          const long long ob = ob0 + g - S4_WARM / 8;
          if (lane < 32 && g >= lo && g < hi && ob >= ob0 && ob < ob1) out[ob] = (uint8_t)byte;
        }
      }
    }
  }
}

}  // namespace dvbt
