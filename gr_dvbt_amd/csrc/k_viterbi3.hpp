// k_viterbi3.hpp -- A7 (viterbi_decoder): depuncture + K=7 Viterbi + traceback, the dominant kernel of the chain.
// Add-compare-select butterflies on DPP lane exchanges (no LDS crossbar, no ds_bpermute in the inner loop), TWO cells
// per VGPR on packed 16-bit arithmetic (v_pk_add_i16 / v_pk_sub_i16 / v_pk_max_i16).
//
// Mapping: one wavefront decodes FOUR chunks, a chunk owns one DPP row (16 lanes); every lane holds 4 of the 64 path
// metrics as the four 16-bit halves of two VGPRs.  The trellis is updated IN PLACE: the butterfly (i, i+32) -> (2i, 2i+1)
// is computed by the two cells that hold states i and i+32, each keeping one output, so after a step the cell that held
// state s holds state rotl6(s): cell c holds state rotl6(c, u mod 6) at relative step u.  The two cells of a butterfly
// differ in exactly one bit of the cell index, which walks 5,4,3,2,1,0,5,... with the step.  Cell index
//     c = r(VGPR 0/1) : h(half 0/1) : a3 a2 a1 a0,   physical lane-in-row = (a0 + 2*a1) ^ (7*a2) ^ (8*a3)
// so the six exchanges are: VGPR swap (free), half swap (one v_alignbit), and four DPP controls (row_ror:8,
// row_half_mirror, quad_perm[2,3,0,1], quad_perm[1,0,3,2]).
//
// Branch metrics enter as delta = 2*(agreements - disagreements) of the butterfly's label with the received pair:
// X = M + delta (own), Y = M - delta (offered to the partner), new = max(X, Y_partner); this is the reference's m0..m3
// (d_viterbi.c:503-506) up to an offset common to all 64 states, which the renormalisation removes anyway.
//
// A 16-bit cell = (2M + bias) << 8 | path byte.
//  * metric field (8 bits, signed): M in units of half an agreement, bias = 1 when the cell holds an upper state
//    (i+32): compares are strict and reproduce the reference's tie rule (d_viterbi.c:508-521).  For a K=7 code
//    the spread of the 64 metrics is bounded by 6 steps x the per-step range: 2M+bias spans <= 49; it drifts by at
//    most +-4 per step, and every second window the best metric is set back to 48 (the reference subtracts the
//    minimum at every output only to keep ITS 8-bit metrics from wrapping, :728-732; any common offset is
//    equivalent), so the field stays within [-65, 113].
//  * path byte (8 bits): the content of the reference's path byte of the survivor -- the state at the window start
//    (kept as that state's cell storage index, bits 5:0) and the two oldest inputs of the window (= top two bits of
//    the state after step 6, stamped there, bits 7:6).  It rides along with the metric through v_pk_max (metric
//    fields never tie).  In the LDS ring a hop is one v_and_or: origin = byte & 63, merged with the next row address.
//  * the tie-break bits are armed THREE STEPS AT A TIME.  Whether a cell holds an upper state at a step depends on one bit of its
//    index, a different one at every step, and the two cells of a butterfly differ in exactly the bit of the current step.  So the
//    biases of the next three steps can be written together into bits 6, 7, 8 (the earliest step in the lowest bit; bits 7:6 of
//    the path byte are free until the stamp of the window's 6th step): at the first step the two candidates agree in bits 8 and 7
//    and differ in bit 6, whichever wins carries the right bits 8 and 7 into the next step, where bit 7 decides and the stale
//    bit 6 below it cannot; and so on.  The deltas have zeros in bits 8:0, so the adds leave the three bits alone.  A window of 8
//    steps re-arms after steps 3, 6 (with the stamp), 7 and 8 (with the origin of the next window): 4 v_and_or instead of 8.
// Per step and VGPR (two cells): v_perm (both branch-metric deltas from ONE word of four class deltas), pk_add,
// pk_sub, exchange, pk_max, and on every second step a v_and_or: 2.75 instructions per cell.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include <string.h>
#include "dvbt_tables.hpp"
#include "k_backend.hpp"

namespace dvbt {

// host side: the decoder's parameters for one configuration (viterbi_decoder_impl.cc:61-65,95-124,149-153)
inline VitParams make_vit_params(const Dims &d, int bsize, int chunk_bytes)
{
  VitParams v; memset(&v, 0, sizeof v);
  v.m = d.m; v.k = d.k; v.n = d.n; v.plen = d.plen; v.ntb = d.ntb; v.bsize = bsize;
  v.d_nsymbols = bsize * d.n / d.m; v.d_nbits = 2 * d.k * bsize;
  v.chunk_bytes = chunk_bytes > 0 ? chunk_bytes : 768; v.payload = d.payload; v.warm = 72;   // (= V3_WARM, declared below)
  memcpy(v.punct, d.punct, 16); memcpy(v.prefix, d.prefix, 16);
  v.punct_mask = 0; v.prefix_nib = 0;
  for (int i = 0; i < d.plen; i++) { v.punct_mask |= (unsigned)d.punct[i] << i; v.prefix_nib |= (unsigned long long)d.prefix[i] << (4 * i); }
  v.magic_plen = ~0ull / (unsigned)d.plen + 1; v.magic_m = ~0ull / (unsigned)d.m + 1;
  for (int i = 0; i < 64; i++) v.punct_rep |= (unsigned long long)d.punct[i % d.plen] << i;
  v.magic16_plen = (65536u + (unsigned)d.plen - 1) / (unsigned)d.plen;
  return v;
}

#define DPP_XOR1 0xB1              /* quad_perm [1,0,3,2] */
#define DPP_XOR2 0x4E              /* quad_perm [2,3,0,1] */
#define DPP_HALF_MIRROR 0x141      /* lane i <-> 7-i inside each half row: logical bit a2 */
#define DPP_ROR8 0x128             /* row_ror:8: lane i <-> i^8 */
#define DPP_MIRROR 0x140

__device__ __forceinline__ int rotl6(int c, int p) { return ((c << p) | (c >> (6 - p))) & 63; }

#ifndef V3_WGW_N
#define V3_WGW_N 4
#endif
constexpr int V3_WGW = V3_WGW_N;    // wavefronts per workgroup: independent (no barrier), each with its own slice of the LDS arrays.  Alone on the machine one-wave and
                                   // four-wave workgroups decode equally fast (3.63 / 3.70 ms); four waves make the workgroup 79 KB of LDS, the size of the symbol kernel's
                                   // (76 KB), so that with several segments in flight a retiring workgroup of either kernel makes room for one of the other and the two
                                   // kernels share the CUs instead of queueing: 4.42 against 4.53 ms per step (bench.py --pipeline 3)
#ifndef V3_EXP
#define V3_EXP 0                   // tools/vit_kbench.hip only, never the product library (cost attribution; the output is wrong for bits 1..8, 256):
                                   // 1 no traceback, 2 no window end, 4 no path-byte store, 8 stage once, 16 record every wavefront's SIMD and
                                   // lifetime, 32 alternate s_setprio window by window, 64 s_nop after every step, 128 s_sleep per window,
                                   // 256 step words from registers (no LDS read in the loop)
#endif
constexpr int V3_WARM = 72;        // warm-up windows before a chunk's first byte (the default; dvbt_rx_params.viterbi_warm_windows selects the WARM = 0 instantiation, which reads VitParams.warm)
constexpr int V3_WARM_MAX = 1152;  // the largest warm-up the parameter accepts
static_assert(V3_WARM == 72, "make_vit_params (above) presets VitParams.warm to the default");
constexpr int V3_BLK = 24;         // windows per forward block (multiple of 6: phase cycle x renormalisation cadence)
constexpr int V3_RINGW = 64;       // windows kept in the LDS ring (power of two, >= 2*V3_BLK - 12 + max ntraceback - 1: the traceback of a
                                   // block runs during the first 12 windows of the next one).  20 KB of LDS per wavefront = 2 wavefronts
                                   // per SIMD.  A 32-window ring with traceback groups of six windows (3 wavefronts per SIMD) was built and
                                   // measured in round 2: +14 % instructions for -7 % cycles per instruction, 4.33 ms against 4.03 ms
                                   // (tools/experiments/viterbi_w3_ring32.patch, profiles/r02_viterbi_attribution.json)
constexpr int V3_WAVES_PER_CU = 4 * 2; // wavefronts resident per CU (LDS: 2 per SIMD)
constexpr int V3_CBW = 17;         // words of compacted received bits per decoder and block (192 steps need <= 384 + 23 bits)

typedef short v3pk __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v3pk pk(int x) { return __builtin_bit_cast(v3pk, x); }
__device__ __forceinline__ int ipk(v3pk x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ int pk_add(int a, int b) { return ipk(pk(a) + pk(b)); }
__device__ __forceinline__ int pk_sub(int a, int b) { return ipk(pk(a) - pk(b)); }
__device__ __forceinline__ int pk_max(int a, int b) { return ipk(__builtin_elementwise_max(pk(a), pk(b))); }
__device__ __forceinline__ int pk_min(int a, int b) { return ipk(__builtin_elementwise_min(pk(a), pk(b))); }
// DPP read with bound_ctrl: every control used here reads a valid lane, and with full row/bank masks the compiler
// then needs no "old" value (no extra v_mov before the v_mov_dpp)
template <int CTRL> __device__ __forceinline__ int dppb(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
// a constant the compiler must keep in a VGPR (v_and_or_b32 takes one literal at most)
__device__ __forceinline__ int vconst(int c) { int x; asm("v_mov_b32 %0, %1" : "=v"(x) : "s"(c)); return x; }
// max of a's halves with b's halves SWAPPED (the half exchange of phase 1 rides on the op_sel of the VOP3P encoding)
__device__ __forceinline__ int pk_max_swap(int a, int b) { int r; asm("v_pk_max_i16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b)); return r; }

struct V3Lane {
  unsigned sel[6][2];   // v_perm selectors: [0, d(class of lo cell), 0, d(class of hi cell)] out of the step word
  int bias8[6][2];      // tie-break bit of phase P at bit 8 of both halves (1 = the cell holds an upper state at that phase)
  int kc[3][2];         // (63 - state of the two cells) | (the cell holds a LOWER state) << 8, at window ends (phase 0,2,4)
  int org[2];           // storage index of the two cells, at the path-byte position (bits 5:0)
  int arm_mid[3][2];    // window starting at phase 0,2,4: biases of its steps 3,4,5 at bits 6,7,8              (after the 3rd step)
  int arm_b2[3][2];     // top two bits of the cells' states after the 6th step at bits 7:6 | bias of step 6 at bit 8 (after the 6th step)
  int arm_org[3][2];    // NEXT window starting at phase 0,2,4: org | biases of its steps 0,1,2 at bits 6,7,8     (after the last step)
};

// storage index of a cell in the path-byte table: z = physical lane * 4 + 2r + h (a lane's four bytes are one word)
__device__ __forceinline__ int v3_phys(int a) { const int a2 = (a >> 2) & 1; return (a & 8) | (a2 << 2) | ((a & 3) ^ (a2 ? 3 : 0)); }
__device__ __forceinline__ int v3_log(int p) { const int q = p & 7, a2 = (q >> 2) & 1; return (p & 8) | (a2 << 2) | ((q ^ (a2 ? 7 : 0)) & 3); }
__device__ __forceinline__ int v3_cell_of_z(int z) { return ((z & 2) << 4) | ((z & 1) << 4) | v3_log(z >> 2); }
__device__ __forceinline__ int v3_z_of_cell(int c) { return v3_phys(c & 15) * 4 + ((c >> 5) & 1) * 2 + ((c >> 4) & 1); }

// 1 when cell (r, h, a) holds an upper state (state bit 5 set) at phase P: the state is rotl6(cell, P), its bit 5 is cell bit (5 - P)
__device__ __forceinline__ int v3_upper(int P, int r, int h, int a) { const int c = (r << 5) | (h << 4) | a; return (c >> (5 - P)) & 1; }
__device__ inline void v3_init_lane(int pl, V3Lane &L)
{
  const int a = v3_log(pl);
  for (int r = 0; r < 2; r++) {
    for (int P = 0; P < 6; P++) {
      unsigned idx[2];
      for (int h = 0; h < 2; h++) {
        const int c = (r << 5) | (h << 4) | a, i = rotl6(c, P) & 31;
        const int c0 = ((i >> 2) ^ (i >> 1) ^ i) & 1;                 // parity(2i & 0x4f)
        const int c1 = ((i >> 4) ^ (i >> 2) ^ (i >> 1)) & 1;          // parity(2i & 0x6d)
        idx[h] = (unsigned)(c0 | (c1 << 1));                          // byte of the step word = label class
      }
      L.sel[P][r] = 0x0c | (idx[0] << 8) | (0x0cu << 16) | (idx[1] << 24);
      L.bias8[P][r] = (v3_upper(P, r, 0, a) << 8) | (v3_upper(P, r, 1, a) << 24);
    }
    L.org[r] = (pl * 4 + 2 * r) | ((pl * 4 + 2 * r + 1) << 16);     // bits 5:0 of the path byte
    for (int e = 0; e < 3; e++) {
      const int P0 = 2 * e;
      const int s0 = rotl6((r << 5) | a, P0), s1 = rotl6((r << 5) | 16 | a, P0);
      L.kc[e][r] = ((63 - s0) | ((63 - s1) << 16)) | (L.bias8[P0][r] ^ 0x01000100);
      const int b2 = ((s0 >> 4) << 6) | (((s1 >> 4) << 6) << 16);   // top two bits of the cells' states at phase P0, at bits 7:6 of the path byte
      // a window that starts at phase P0: step u runs at phase (P0 + u) % 6
      L.arm_mid[e][r] = (L.bias8[(P0 + 3) % 6][r] >> 2) | (L.bias8[(P0 + 4) % 6][r] >> 1) | L.bias8[(P0 + 5) % 6][r];
      L.arm_org[e][r] = L.org[r] | (L.bias8[P0][r] >> 2) | (L.bias8[P0 + 1][r] >> 1) | L.bias8[(P0 + 2) % 6][r];
      // after the 6th step of a window that started at phase P0 the phase is P0 again: the stamp and the bias of the 7th step
      L.arm_b2[e][r] = b2 | L.bias8[P0][r];
    }
  }
}

// ST: what happens to the tie-break bits after the step (see the header): 0 nothing (the bits armed earlier serve the next step too);
// 3 = the window's 3rd step: the biases of steps 4..6 are armed; 1 = the 6th step: the two oldest inputs are stamped into the path byte
// together with the bias of the 7th; 4 = the 7th step: the bias of the 8th; 2 = the last step: raw[] keeps the survivors' path bytes for the
// table, v gets the origin stamp of the next window and the biases of its first three steps in the same v_and_or
template <int P, int ST> __device__ __forceinline__ void v3_step(int (&v)[2], unsigned W, const V3Lane &L, int (&raw)[2])
{
  int X[2], Y[2], mx[2];
  // The label class of a cell depends on bits 0,1,2,4 of its state (parity taps 0x4f, 0x6d without the MSB).  The
  // VGPR index is cell bit 5 = state bit (5+P)%6: at phases 0 and 4 (state bits 5, 3) both VGPRs have the same
  // deltas, at phases 2 and 3 (state bits 1, 2: both parities flip) VGPR 1 has the negated ones.
  const int D0 = (int)__builtin_amdgcn_perm(0u, W, L.sel[P][0]);
  X[0] = pk_add(v[0], D0); Y[0] = pk_sub(v[0], D0);
  if (P == 0 || P == 4) { X[1] = pk_add(v[1], D0); Y[1] = pk_sub(v[1], D0); }
  else if (P == 2 || P == 3) { X[1] = pk_sub(v[1], D0); Y[1] = pk_add(v[1], D0); }
  else {
    const int D1 = (int)__builtin_amdgcn_perm(0u, W, L.sel[P][1]);
    X[1] = pk_add(v[1], D1); Y[1] = pk_sub(v[1], D1);
  }
  if (P == 0) { mx[0] = pk_max(X[0], Y[1]); mx[1] = pk_max(X[1], Y[0]); }
  else if (P == 1) { mx[0] = pk_max_swap(X[0], Y[0]); mx[1] = pk_max_swap(X[1], Y[1]); }
  else {
#pragma unroll
    for (int r = 0; r < 2; r++)
      mx[r] = pk_max(X[r], P == 2 ? dppb<DPP_ROR8>(Y[r]) : P == 3 ? dppb<DPP_HALF_MIRROR>(Y[r]) : P == 4 ? dppb<DPP_XOR2>(Y[r]) : dppb<DPP_XOR1>(Y[r]));
  }
  constexpr int PN = (P + 1) % 6;
#pragma unroll
  for (int r = 0; r < 2; r++) {
    if (ST == 0) v[r] = mx[r];
    else if (ST == 3) v[r] = (mx[r] & (int)0xfe3ffe3f) | L.arm_mid[((PN + 3) % 6) / 2][r];       // the window started at phase PN - 3
    else if (ST == 1) v[r] = (mx[r] & (int)0xfe3ffe3f) | L.arm_b2[PN / 2][r];                    // PN = the phase the window started at
    else if (ST == 4) v[r] = (mx[r] & (int)0xfefffeff) | L.bias8[PN][r];
    else { raw[r] = mx[r]; v[r] = (mx[r] & (int)0xfe00fe00) | L.arm_org[PN / 2][r]; }
  }
}

// the 8 steps of a window that starts at phase P0 (0, 2, 4); v arrives with the origin stamp in its path bytes
#if V3_EXP & 64
#define V3_YIELD() asm volatile("s_nop 3" ::: "memory")
#else
#define V3_YIELD()
#endif
template <int P0> __device__ __forceinline__ void v3_window(int (&v)[2], const unsigned (&W)[8], const V3Lane &L, int (&raw)[2])
{
  v3_step<(P0 + 0) % 6, 0>(v, W[0], L, raw); V3_YIELD(); v3_step<(P0 + 1) % 6, 0>(v, W[1], L, raw); V3_YIELD(); v3_step<(P0 + 2) % 6, 3>(v, W[2], L, raw); V3_YIELD();
  v3_step<(P0 + 3) % 6, 0>(v, W[3], L, raw); V3_YIELD(); v3_step<(P0 + 4) % 6, 0>(v, W[4], L, raw); V3_YIELD(); v3_step<(P0 + 5) % 6, 1>(v, W[5], L, raw); V3_YIELD();
  v3_step<(P0 + 6) % 6, 4>(v, W[6], L, raw); V3_YIELD(); v3_step<(P0 + 7) % 6, 2>(v, W[7], L, raw);
}

// halves of a packed register as sign-extended 32-bit values
__device__ __forceinline__ int lo16(int x) { return (x << 16) >> 16; }
__device__ __forceinline__ int hi16(int x) { return x >> 16; }
template <int CTRL> __device__ __forceinline__ int max_dpp(int v) { return max(v, dppb<CTRL>(v)); }   // folds into v_max_i32_dpp
template <int CTRL> __device__ __forceinline__ int min_dpp(int v) { return min(v, dppb<CTRL>(v)); }

// end of a window: best state = first index of the maximum metric (d_viterbi.c:699-711) and, when asked, the
// renormalisation.  PE = phase after the window.  Key of a cell = (2M + bias ^ 1) << 8 | 63 - state: among equal M the
// cells holding a lower state win over the upper ones, then the smaller state (the flag sits in L.kc); one v_and_or per VGPR.
// Returns the key's low byte (63 - best state in bits 5:0) in every lane of the row.
template <int PE, bool RENORM> __device__ __forceinline__ int v3_window_end(int (&v)[2], const V3Lane &L)
{
  const int kp = pk_max((v[0] & (int)0xfe00fe00) | L.kc[PE / 2][0], (v[1] & (int)0xfe00fe00) | L.kc[PE / 2][1]);
  int k = max(lo16(kp), hi16(kp));
  k = max_dpp<DPP_XOR1>(k); k = max_dpp<DPP_XOR2>(k); k = max_dpp<DPP_HALF_MIRROR>(k); k = max_dpp<DPP_MIRROR>(k);
  if (RENORM) {
    // any offset common to the 64 metrics will do (the reference subtracts the minimum, d_viterbi.c:728-732, only to keep
    // its 8-bit metrics from wrapping): the maximum is already here, so the best metric is set to 2*24 -- the spread is
    // at most 49, so the field restarts inside [-1, 49] and drifts by at most +-64 until the next renormalisation
    const unsigned sub = (unsigned)((k & ~0x1ff) - (48 << 8));        // (2 M_best - 48) at the metric position; bias and path byte untouched
    const int mn = (int)__builtin_amdgcn_perm(sub, sub, 0x01000100u);
    v[0] = pk_sub(v[0], mn); v[1] = pk_sub(v[1], mn);
  }
  return k;
}

// Two traceback chains per lane: the calls (windows) pl and 16+pl of one block of a decoder.
struct V3Trace {
  int z[2];             // current cell (storage index)
  int wsh[2];           // ring row of the window whose table is read next (<< 8, counts down past zero: only bits 13:8 are used) | decoder row << 6
  int wlast[2];         // the window a chain ends in (relative index)
  bool ok[2];
  long long ob[2];      // output byte of the call
};
// one hop of both chains: state = path_byte >> 2 in the reference's layout (d_viterbi.c:717) = the low six bits here.
// wa = ring row of the window to read | decoder row; the v_and_or that extracts the origin also forms the address.
__device__ __forceinline__ void v3_hop(V3Trace &T, const unsigned char *tab, int rowc)
{
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const unsigned t = tab[T.z[q]];                                  // z holds the full LDS index
    T.wsh[q] -= 256;                                                 // one ring row back; the decoder row rides in bits 7:6, the wrap is the mask below
    T.z[q] = (T.wsh[q] & 0x3fc0) | ((int)t & ~0x3fc0);               // one v_bfi: ring row and decoder row from wsh, the origin (bits 5:0) from the path byte
  }
}
// the decoded byte of a call: (state at the start of the last window of the chain) << 2 | its two oldest inputs.
// w = that window (relative index, for its phase)
__device__ __forceinline__ void v3_trace_out(const V3Trace &T, const unsigned char *tab, int rowc, uint8_t *out, long long out_lo)
{
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const unsigned t = tab[T.z[q]];
    const int w = T.wlast[q];
    const int sstart = rotl6(v3_cell_of_z((int)(t & 63u)), 2 * (((w % 3) + 3) % 3));    // phase of window w = (8w) % 6
    if (T.ok[q]) out[T.ob[q] - out_lo] = (unsigned char)((sstart << 2) | (t >> 6));
  }
}

// window j0 + V6 of a decoder (j0 % 6 == 0): it starts at phase (8 V6) % 6 = 0,2,4,0,2,4; the minimum is subtracted
// after every second window.  HOPS: two traceback hops of the previous block's calls ride along (their LDS latency
// hides under the add-compare-select work).
// Which hops exist is a compile-time fact (HOP0 + 2 V6 (+1) < NTB - 1 with NTB = ntraceback, a template parameter of the
// kernel): the window stays one straight-line block and the scheduler can spread the dependent LDS reads over it.
template <int V6, bool HOPS, int HOP0, int NTB> __device__ __forceinline__ void v3_fwd_window(int (&v)[2], const V3Lane &L, const unsigned *wrow, unsigned char *tab,
                                                                                           unsigned char *bests, int j0, int dd, int pl, V3Trace &T)
{
  if (HOPS && HOP0 + 2 * V6 < NTB - 1) v3_hop(T, tab, dd * 64);
  if (HOPS && HOP0 + 2 * V6 + 1 < NTB - 1) v3_hop(T, tab, dd * 64);
  const int jr = (j0 + V6) & (V3_RINGW - 1);
  unsigned W[8];
#if V3_EXP & 256
  for (int i = 0; i < 8; i++) { W[i] = (unsigned)(j0 * 0x01010101 + i * 0x00020406 + pl); asm volatile("" : "+v"(W[i])); }   // no LDS read in the loop
#else
  {
    const uint4 *wp = reinterpret_cast<const uint4 *>(wrow + V6 * 8);
    const uint4 t0 = wp[0], t1 = wp[1];
    W[0] = t0.x; W[1] = t0.y; W[2] = t0.z; W[3] = t0.w; W[4] = t1.x; W[5] = t1.y; W[6] = t1.z; W[7] = t1.w;
  }
#endif
  constexpr int P0 = (8 * V6) % 6;
  int raw[2];
#if V3_EXP & 32
  if (V6 & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);   // alternate the wavefront's issue priority window by window
#endif
#if V3_EXP & 128
  __builtin_amdgcn_s_sleep(1);
#endif
  v3_window<P0>(v, W, L, raw);
  // the four path bytes of this lane's cells = one word of the table (storage index z = 4*lane + 2r + h)
  if (!(V3_EXP & 4)) *reinterpret_cast<unsigned *>(tab + (jr * 4 + dd) * 64 + pl * 4) = __builtin_amdgcn_perm((unsigned)raw[1], (unsigned)raw[0], 0x06040200u);
  if (V3_EXP & 2) { if (raw[0] == 0x12345) bests[jr] = 1; return; }
  const int s = v3_window_end<(P0 + 2) % 6, (V6 & 1) == 1>(v, L);
  bests[dd * V3_RINGW + jr] = (unsigned char)s;                    // low byte of the key (63 - best state in bits 5:0); all 16 lanes of the row write the same byte
}
template <bool HOPS, int HOP0, int NTB> __device__ __forceinline__ void v3_fwd_six(int (&v)[2], const V3Lane &L, const unsigned *wrow, unsigned char *tab,
                                                                                  unsigned char *bests, int j0, int dd, int pl, V3Trace &T)
{
  v3_fwd_window<0, HOPS, HOP0, NTB>(v, L, wrow, tab, bests, j0, dd, pl, T); v3_fwd_window<1, HOPS, HOP0, NTB>(v, L, wrow, tab, bests, j0, dd, pl, T);
  v3_fwd_window<2, HOPS, HOP0, NTB>(v, L, wrow, tab, bests, j0, dd, pl, T); v3_fwd_window<3, HOPS, HOP0, NTB>(v, L, wrow, tab, bests, j0, dd, pl, T);
  v3_fwd_window<4, HOPS, HOP0, NTB>(v, L, wrow, tab, bests, j0, dd, pl, T); v3_fwd_window<5, HOPS, HOP0, NTB>(v, L, wrow, tab, bests, j0, dd, pl, T);
}

// A chunk = vp.chunk_bytes decoded bytes; it is decoded by an independent decoder that starts `warm` windows early
// from all-zero metrics and runs ntraceback-1 windows past its end.  Per block of 24 windows:
//   staging   depuncture (viterbi_decoder_impl.cc:241-256) + delta packing for 192 steps x 4 decoders.  The row
//             loads the input bytes of its decoder (one 16-byte load per lane, issued one block ahead) and compacts
//             the m valid bits of every byte into a bit stream in LDS; every lane then owns 12 consecutive steps:
//             it places itself in the puncture period with small-integer arithmetic relative to the row's base (one
//             64-bit locate per row), takes the period's keep-mask for its 24 symbol positions and a 32-bit window
//             of the bit stream, and emits one word of four class deltas per step through a 16-entry table;
//   forward   24 windows of add-compare-select (v3_fwd_window), path bytes into the LDS ring;
//   traceback of the PREVIOUS block's 24 calls x 4 decoders (d_viterbi.c:714-724), lane = (decoder, call): its
//             dependent LDS reads are interleaved into the first 12 windows of the forward pass.
#if V3_EXP & 16
__device__ unsigned long long *v3_dbg;     // tools/vit_kbench.hip: (hw id, start, end in 100 MHz ticks) per wavefront
#endif

// ---- The chunk decoders ARE the reference's one streaming decoder (viterbi_decoder_impl.cc:192-324, d_viterbi.c:680-735), by construction: proof + repair.
// A decoder's whole state at the top of a block of windows is its two registers of cells: the block starts at phase 0 right behind a renormalisation (best metric = 48) and the
// low nine bits of a cell are a function of the lane and the phase alone (v3_step, ST == 2) -- two decoders over the same input whose registers are EQUAL there make identical
// decisions from there on.  With B a multiple of V3_BLK the decoder of chunk c stands at the top of a block when it reaches its chunk's first window (relative window warm) and
// so does its predecessor, B windows further into its own run: both store their registers (MODE 1).  Per chunk c three slots of 32 words:
//   own[c]   the state, at the chunk's first window, of the decoder whose bytes fill chunk c
//   pred[c]  the state there of the decoder whose bytes fill chunk c - 1
//   fix[c]   (repair) the state there of a repaired chunk c - 1 where it is not pred[c]
// Chunk 0 starts the stream (or is handed the streaming decoder's state: the single block carries it from call to call), so own[c] == pred[c] for every c >= 1 proves, by induction,
// that every byte is the streaming decoder's.  viterbi_check_kernel lists the chunks where the two differ (their survivors had not merged inside the warm-up: none in 83,000 on
// streams the code can cope with, one in a few hundred on a collapsed channel or the hierarchical modes' degenerate input); viterbi_repair_kernel decodes each of them again from
// pred[c] (MODE 2: no warm-up, one chunk, in parallel) and compares the state it reaches at the next chunk's first window with pred[c + 1] -- equal: the chain holds (whoever
// filled chunk c + 1 started from, or was checked against, that state).  Not equal (the repaired decoder and the unproven one have not merged over a whole chunk: never observed)
// the chunk behind is flagged and viterbi_repair_seq_kernel, one decoder, walks on from there in stream order until its state is the own[] of the chunk it arrives at.  The host
// reads no verdict: the launches are unconditional and return at once when there is nothing to do.
constexpr int V3_SLOT = 32;          // words per state slot (16 lanes x 2 registers)
constexpr int V3_CTL_CHUNKS = 0;     // ctl words: chunks of the launch
constexpr int V3_CTL_MISMATCH = 1;   //            chunks the first check could not prove (= decoded again)
constexpr int V3_CTL_CONFLICT = 2;   //            chunks left to the sequential pass
constexpr int V3_CTL_UNPROVEN = 3;   //            chunks that are not proven after the repair (the final check; -1: not run)
constexpr int V3_CTL_SEQ = 4;        //            chunks the sequential pass decoded
constexpr int V3_CTL_ACC = 8;        //            [8..10] chunks, mismatches, sequential decodes summed over the launches since the owner cleared them
constexpr int V3_CTL_HDR = 16;       // ... then the list of mismatching chunks (cap words), then the conflict flags (cap words)
struct V3Aux {
  int *snap;                         // 3 slots per chunk (own, pred, fix), cap + 2 chunks
  int *ctl;
  long long cap;                     // chunks the buffers hold
  long long grid0;                   // absolute index of chunk 0's first byte (the single block: where the carried state stands, <= the first byte of the call)
  const int *carry_in;               // the streaming decoder's state at grid0 (null: chunk 0 warms up like the others -- it starts the stream)
  int *carry_out; int carry_rel;     // the last chunk's decoder leaves its state carry_rel windows into its chunk (a multiple of V3_BLK; carry_out null: no)
  int debug;                         // test hook (dvbt_rx_params.viterbi_verify = 4): bit 0 = every repaired chunk flags the chunk behind it as if its decoder had not arrived in
                                     // pred[] (the flag / fix[] hand-over to the sequential pass is otherwise reached only by inputs nobody has seen)
};
__device__ __forceinline__ int *v3_own(const V3Aux &ax, long long c) { return ax.snap + (3 * c) * V3_SLOT; }
__device__ __forceinline__ int *v3_pred(const V3Aux &ax, long long c) { return ax.snap + (3 * c + 1) * V3_SLOT; }
__device__ __forceinline__ int *v3_fix(const V3Aux &ax, long long c) { return ax.snap + (3 * c + 2) * V3_SLOT; }
inline size_t v3_snap_words(long long cap) { return (size_t)(cap + 2) * 3 * V3_SLOT; }
inline size_t v3_ctl_words(long long cap) { return (size_t)V3_CTL_HDR + 2 * (size_t)(cap + 2); }

struct V3Lds { unsigned char *tab; unsigned *wbuf; unsigned char *bests; const unsigned *lut; };
// the label-class deltas of a step as a function of (keep flags, next two received bits); every wavefront writes the same 16 words
__device__ __forceinline__ void v3_init_lut(unsigned *lut, int lane)
{
  if (lane < 16) {
    // deltas of the four label classes, doubled (a step with one punctured symbol keeps the bias bit free):
    // class 0: u0+u1 | class 1 (c0=1): -u0+u1 | class 2 (c1=1): u0-u1 | class 3: -u0-u1, u = +1/-1 for a received 0/1, 0 if erased
    const int k0 = (lane >> 2) & 1, k1 = (lane >> 3) & 1, t1 = (lane >> 1) & 1, t0 = lane & 1;
    const int u0 = k0 ? 1 - 2 * t1 : 0, u1 = k1 ? 1 - 2 * (k0 ? t0 : t1) : 0;
    const unsigned d0 = (unsigned)(2 * (u0 + u1)) & 0xff, d1 = (unsigned)(2 * (-u0 + u1)) & 0xff;
    const unsigned d2 = (unsigned)(2 * (u0 - u1)) & 0xff, d3 = (unsigned)(2 * (-u0 - u1)) & 0xff;
    lut[lane] = d0 | (d1 << 8) | (d2 << 16) | (d3 << 24);
  }
}

// One decoder per DPP row over [b0, min(b0 + B, total_out)), the wavefront's four rows in step.
// MODE 0: the plain chunk decoder.  MODE 1: the same, and it leaves own[] / pred[] (own, predn: this lane's two words of the slots; inject: the state chunk 0 is handed at its
// first window instead of what its warm-up produced).  MODE 2: from the state in v, no warm-up (WARM is ignored); endv = the state at the next chunk's first window.
// WARM: the warm-up in windows as a compile-time constant (the default instantiation: V3_WARM), or 0: taken from vp.warm (any multiple of V3_BLK up to V3_WARM_MAX).
template <int NTB, int WARM, int MODE>
__device__ __forceinline__ void v3_decode(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, long long total_steps, long long total_out, const VitParams &vp,
                                          long long in_base, long long out_lo, long long b0, bool dec_active, const V3Lds &S, const V3Lane &L, int (&v)[2], int (&endv)[2],
                                          int *own, int *predn, const int *inject, int *carry, int carry_rel)
{
  unsigned char *const tab = S.tab; unsigned *const wbuf = S.wbuf; unsigned char *const bests = S.bests; const unsigned *const lut = S.lut;
  // compacted received bits per decoder (MSB first): V3_CBW words at the head of the decoder's wbuf row.  stage_words reads them (every
  // lane its two words) before the wavefront writes the block's step words over them, and the previous block's step words are dead by
  // then; a wavefront's LDS operations execute in order.  The overlay keeps the workgroup at 19,776 B of LDS per wavefront: eight workgroups per CU
  // (two wavefronts per SIMD) with room to spare
  unsigned *const cbits = wbuf;
  constexpr int V3_CBS = V3_BLK * 8;                                               // stride between the decoders' bit streams
  static_assert(V3_CBW <= V3_CBS, "bit stream does not fit the step-word row");
  const int lane = threadIdx.x & 63, dd = lane >> 4, pl = lane & 15;
  const int B = vp.chunk_bytes, m = vp.m;
  constexpr int ntb = NTB;                                         // == vp.ntb (the host picks the instantiation)
  const long long b1 = (b0 + B < total_out) ? b0 + B : total_out;
  const int warm = MODE == 2 ? 0 : WARM > 0 ? WARM : vp.warm;
  const long long w0 = b0 + 2 - warm;                              // absolute window of relative window 0
  const int J = ((warm + B + ntb - 1 + V3_BLK - 1) / V3_BLK) * V3_BLK;
  const long long n_in_bytes = (total_steps * 2 / vp.plen * vp.n + vp.m - 1) / vp.m;   // input bytes that exist
  const int nload = ((384 + 2 * m - 2) / m + 3 + 15) / 16;         // lanes whose 16 bytes a block can need (13, 7, 5)

  // ---- staging, first half: where block jb starts in the input (row-uniform) and the load of its bytes
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
    const long long src = byte0 - off + pl * 16;                   // 4-byte aligned address
    q = make_uint4(0, 0, 0, 0);
    if (dec_active && pl < nload) {
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
  // ---- staging, second half: bit compaction and the step words of block jb
  auto stage_words = [&](int jb) {
    {
      const unsigned mk = (1u << m) - 1;
      auto grp = [&](unsigned d) { return ((d & mk) << (3 * m)) | (((d >> 8) & mk) << (2 * m)) | (((d >> 16) & mk) << m) | ((d >> 24) & mk); };
      const unsigned g0 = grp(q.x), g1 = grp(q.y), g2 = grp(q.z), g3 = grp(q.w);
      unsigned *cb = cbits + dd * V3_CBS;
      if (pl < nload) {
        if (m == 2) cb[pl] = (g0 << 24) | (g1 << 16) | (g2 << 8) | g3;
        else if (m == 4) { cb[2 * pl] = (g0 << 16) | g1; cb[2 * pl + 1] = (g2 << 16) | g3; }
        else { cb[3 * pl] = (g0 << 8) | (g1 >> 16); cb[3 * pl + 1] = (g1 << 16) | (g2 >> 8); cb[3 * pl + 2] = (g2 << 24) | g3; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    constexpr int SPL = V3_BLK * 8 / 16;                           // steps per lane
    const long long tb = 8 * (w0 - 1) + (long long)jb * 8 - 2;
    const long long tbr = tb > 0 ? tb : 0;
    const int ub0 = pl * SPL;
    const long long t = tb + ub0;
    int lo = 0;                                                    // leading steps before the stream start
    if (t < 0) lo = (-t < SPL) ? (int)(-t) : SPL;
    const long long rem = total_steps - t;
    int hi = !dec_active ? 0 : rem <= 0 ? 0 : rem < SPL ? (int)rem : SPL;
    if (hi < lo) hi = lo;
    int x = ph0 + 2 * (int)((t + lo) - tbr);                        // depunctured-bit offset of the first real step from the row base
    if (x < 0) x = 0;
    const int dq = (int)(((unsigned)x * vp.magic16_plen) >> 16);
    const int ph = x - dq * vp.plen;
    const int pos = off * m + bo0 + dq * vp.n + (int)((vp.prefix_nib >> (4 * ph)) & 15ull) - (int)((vp.prefix_nib >> (4 * ph0)) & 15ull);
    const unsigned kmask = ((unsigned)(vp.punct_rep >> ph) << (2 * lo)) & (((1u << (2 * hi)) - 1u) & ~((1u << (2 * lo)) - 1u));
    const unsigned *cb = cbits + dd * V3_CBS;
    const unsigned cw0 = cb[pos >> 5], cw1 = cb[(pos >> 5) + 1];
    unsigned win = (unsigned)(((((unsigned long long)cw0) << 32) | cw1) >> (32 - (pos & 31)));
#pragma unroll
    for (int i = 0; i < SPL; i++) {
      const unsigned k2 = (kmask >> (2 * i)) & 3u;                  // keep flags of the step's two symbols
      const unsigned idx = (k2 << 2) | (win >> 30);
      win <<= (k2 - (k2 >> 1));                                    // consume one received bit per kept symbol
      wbuf[dd * (V3_BLK * 8) + ub0 + i] = lut[idx];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  };
  // ---- traceback chains of the block that starts at window jp
  V3Trace T;
  auto trace_init = [&](int jp) {
#pragma unroll
    for (int c = 0; c < 2; c++) {
      int jj = jp + c * 16 + pl;
      T.ob[c] = b0 + (jj - (warm + ntb - 1));
      T.ok[c] = (c * 16 + pl < V3_BLK) && dec_active && jj >= warm + ntb - 1 && T.ob[c] < b1 && (MODE != 1 || T.ob[c] >= out_lo);
      if (!T.ok[c]) jj = jp;                                       // any window inside the ring: result unused
      const int sb = 63 - (bests[dd * V3_RINGW + (jj & (V3_RINGW - 1))] & 63);
      T.wsh[c] = ((jj << 8) & 0x3f00) | (dd * 64);
      T.z[c] = v3_z_of_cell(((sb | (sb << 6)) >> ((8 * jj + 8) % 6)) & 63) | T.wsh[c];   // cell = rotr6(state, phase after the window)
      T.wlast[c] = jj - (ntb - 1);
    }
  };

  stage_load(0);
  for (int jb = 0; jb < J; jb += V3_BLK) {
    if (MODE == 1 && dec_active) {                                 // (wave-uniform but for dec_active and the pointers)
      if (jb == warm) { if (inject) { v[0] = inject[0]; v[1] = inject[1]; } own[0] = v[0]; own[1] = v[1]; }
      if (jb == warm + B) { predn[0] = v[0]; predn[1] = v[1]; }   // the NEXT chunk's first window, seen by its predecessor
      if (carry && jb == warm + carry_rel) { carry[0] = v[0]; carry[1] = v[1]; }
    }
    if (MODE == 2) {
      if (jb == 0 && dec_active) { own[0] = v[0]; own[1] = v[1]; }
      if (jb == B) { endv[0] = v[0]; endv[1] = v[1]; }
      if (carry && dec_active && jb == carry_rel) { carry[0] = v[0]; carry[1] = v[1]; }
    }
    if (!(V3_EXP & 8) || jb == 0) stage_words(jb);
    if (jb + V3_BLK < J && !(V3_EXP & 8)) stage_load(jb + V3_BLK);                  // the bytes of the next block travel during this block's forward pass
    const bool tr = jb > 0 && !(V3_EXP & 1);
    if (tr) trace_init(jb - V3_BLK);
    const unsigned *wrow = wbuf + dd * (V3_BLK * 8);
    if (tr) {
      v3_fwd_six<true, 0, NTB>(v, L, wrow, tab, bests, jb, dd, pl, T);
      v3_fwd_six<true, 12, NTB>(v, L, wrow + 48, tab, bests, jb + 6, dd, pl, T);
      v3_trace_out(T, tab, dd * 64, out, out_lo);
    } else {
      v3_fwd_six<false, 0, NTB>(v, L, wrow, tab, bests, jb, dd, pl, T);
      v3_fwd_six<false, 0, NTB>(v, L, wrow + 48, tab, bests, jb + 6, dd, pl, T);
    }
    for (int wi = 12; wi < V3_BLK; wi += 6) v3_fwd_six<false, 0, NTB>(v, L, wrow + wi * 8, tab, bests, jb + wi, dd, pl, T);
  }
  // the last block's calls
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  trace_init(J - V3_BLK);
  for (int h = 0; h < ntb - 1; h++) v3_hop(T, tab, dd * 64);
  v3_trace_out(T, tab, dd * 64, out, out_lo);
}

#define V3_LDS_DECL(NW) \
  __shared__ __attribute__((aligned(16))) unsigned char tab_[NW][V3_RINGW * 4 * 64];   /* path bytes: [window][decoder][cell z]  (the ppresult ring) */ \
  __shared__ __attribute__((aligned(16))) unsigned wbuf_[NW][4 * V3_BLK * 8];          /* step words: [decoder][step in block] */ \
  __shared__ unsigned char bests_[NW][4 * V3_RINGW];                                    /* best state per window */ \
  __shared__ unsigned lut[16];                                                          /* (keep flags, next two received bits) -> the step word */

// MODE 0 / 1 (see v3_decode).  MODE 0: out_lo is also the index of chunk 0's first byte (the block API's former layout); MODE 1: ax.grid0 is, out_lo the first byte that is written.
template <int NTB, int WARM = V3_WARM, int MODE = 0> __global__ __launch_bounds__(64 * V3_WGW) void viterbi3_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const RxState *st,
                                                      long long steps_fixed, VitParams vp, V3Aux ax, long long in_base, long long out_lo)
{
  V3_LDS_DECL(V3_WGW)
  const int wv = V3_WGW > 1 ? (int)(threadIdx.x >> 6) : 0;
  const V3Lds S = {tab_[wv], wbuf_[wv], bests_[wv], lut};
  const int lane = threadIdx.x & 63, dd = lane >> 4, pl = lane & 15;
  const long long total_steps = st ? st->n_vit_steps : steps_fixed;
  const long long total_out = total_steps / 8 - vp.ntb;
  const int B = vp.chunk_bytes;
  if (MODE == 1 && blockIdx.x == 0 && threadIdx.x == 0) {        // the counters of the passes behind this launch
    ax.ctl[V3_CTL_MISMATCH] = 0; ax.ctl[V3_CTL_CONFLICT] = 0; ax.ctl[V3_CTL_UNPROVEN] = -1; ax.ctl[V3_CTL_SEQ] = 0;
  }
  const long long grid0 = MODE == 1 ? ax.grid0 : out_lo;
  const long long chunk0 = ((long long)blockIdx.x * V3_WGW + wv) * 4;
  if (grid0 + chunk0 * B >= total_out) return;                    // whole wavefront idle
  const long long c = chunk0 + dd, b0 = grid0 + c * B;            // this lane's decoder
  const bool dec_active = b0 < total_out;
  v3_init_lut(lut, lane);
#if V3_EXP & 16
  const unsigned long long dbg_t0 = wall_clock64();
#endif
  V3Lane L; v3_init_lane(pl, L);
  int v[2] = {L.arm_org[0][0], L.arm_org[0][1]}, endv[2];          // all-zero metrics, origin stamp of window 0, tie-break bits of its first three steps
  int *own = nullptr, *predn = nullptr, *carry = nullptr; const int *inject = nullptr;
  if (MODE == 1) {
    own = v3_own(ax, c) + 2 * pl; predn = v3_pred(ax, c + 1) + 2 * pl;
    if (c == 0 && ax.carry_in) inject = ax.carry_in + 2 * pl;
    if (ax.carry_out && b0 + B >= total_out) carry = ax.carry_out + 2 * pl;      // the launch's last chunk
  }
  v3_decode<NTB, WARM, MODE>(in, out, total_steps, total_out, vp, in_base, out_lo, b0, dec_active, S, L, v, endv, own, predn, inject, carry, ax.carry_rel);
#if V3_EXP & 16
  if (lane == 0) {
    unsigned id, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long *d = v3_dbg + 3 * ((long long)blockIdx.x * V3_WGW + wv);
    d[0] = (id & 0xffffu) | ((xcc & 0xfu) << 16); d[1] = dbg_t0; d[2] = wall_clock64();
  }
#endif
}

__device__ __forceinline__ long long v3_nchunks(long long total_out, long long grid0, int B) { return total_out > grid0 ? (total_out - grid0 + B - 1) / B : 0; }
__device__ __forceinline__ bool v3_slots_equal(const int *a_, const int *b_)
{
  const int4 *a = reinterpret_cast<const int4 *>(a_), *b = reinterpret_cast<const int4 *>(b_);      // 32 words = 8 x int4 per slot
  bool same = true;
#pragma unroll
  for (int i = 0; i < 8; i++) { const int4 x = a[i], y = b[i]; same = same && x.x == y.x && x.y == y.y && x.z == y.z && x.w == y.w; }
  return same;
}

// pass 0: the chunks c >= 1 whose own[c] is not pred[c] go on the list (ctl[V3_CTL_MISMATCH] of them), the conflict flags are cleared.
// pass 1 (the final check, dvbt_rx_params.viterbi_verify): they are counted in ctl[V3_CTL_UNPROVEN] (zero behind the repair passes, by construction).
__global__ __launch_bounds__(256) void viterbi_check_kernel(V3Aux ax, const RxState *st, long long steps_fixed, VitParams vp, int pass)
{
  const long long total_steps = st ? st->n_vit_steps : steps_fixed;
  const long long nch = v3_nchunks(total_steps / 8 - vp.ntb, ax.grid0, vp.chunk_bytes);
  if (blockIdx.x == 0 && threadIdx.x == 0) { ax.ctl[V3_CTL_CHUNKS] = (int)(nch < INT_MAX ? nch : INT_MAX); if (pass == 1) ax.ctl[V3_CTL_UNPROVEN] = 0; }
  const long long c = (long long)blockIdx.x * 256 + threadIdx.x + 1;
  if (c >= nch || c > ax.cap) return;
  const bool same = v3_slots_equal(v3_own(ax, c), v3_pred(ax, c));
  if (pass == 0) {
    ax.ctl[V3_CTL_HDR + (ax.cap + 2) + c] = 0;
    if (!same) ax.ctl[V3_CTL_HDR + atomicAdd(ax.ctl + V3_CTL_MISMATCH, 1)] = (int)c;
  }
}
// (the final check's count needs its zero before any thread adds to it: a second kernel rather than a grid-wide handshake)
__global__ __launch_bounds__(256) void viterbi_count_kernel(V3Aux ax, const RxState *st, long long steps_fixed, VitParams vp)
{
  const long long total_steps = st ? st->n_vit_steps : steps_fixed;
  const long long nch = v3_nchunks(total_steps / 8 - vp.ntb, ax.grid0, vp.chunk_bytes);
  const long long c = (long long)blockIdx.x * 256 + threadIdx.x + 1;
  if (c >= nch || c > ax.cap) return;
  if (!v3_slots_equal(v3_own(ax, c), v3_pred(ax, c))) atomicAdd(ax.ctl + V3_CTL_UNPROVEN, 1);
}

// every lane of the row: do the row's 16 lanes hold the same two words in a and b?
__device__ __forceinline__ bool v3_row_equal(const int (&a)[2], const int (&b)[2], int dd)
{
  const unsigned long long ne = __ballot(a[0] != b[0] || a[1] != b[1]);
  return ((ne >> (16 * dd)) & 0xffffull) == 0;
}

// The parallel repair pass: one decoder (row) per listed chunk, from pred[c], over that chunk alone.  `rows` decoder rows work through the list, this wavefront's first is row0.
template <int NTB> __device__ __forceinline__ void v3_repair_rows(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, long long total_steps, const VitParams &vp, const V3Aux &ax,
                                                                  long long in_base, long long out_lo, const V3Lds &S, const V3Lane &L, long long rows, long long row0, int n)
{
  const int lane = threadIdx.x & 63, dd = lane >> 4, pl = lane & 15;
  const long long total_out = total_steps / 8 - vp.ntb;
  const int B = vp.chunk_bytes;
  const long long nch = v3_nchunks(total_out, ax.grid0, B);
  for (long long i0 = row0; i0 < n; i0 += rows) {
    const bool act = i0 + dd < n;
    const long long c = act ? ax.ctl[V3_CTL_HDR + i0 + dd] : 1;
    const long long b0 = ax.grid0 + c * B;
    int v[2], endv[2] = {0, 0};
    { const int *p = v3_pred(ax, c) + 2 * pl; v[0] = p[0]; v[1] = p[1]; }
    int *carry = (act && ax.carry_out && b0 + B >= total_out) ? ax.carry_out + 2 * pl : nullptr;
    v3_decode<NTB, 0, 2>(in, out, total_steps, total_out, vp, in_base, out_lo, b0, act && b0 < total_out, S, L, v, endv, v3_own(ax, c) + 2 * pl, nullptr, nullptr, carry, ax.carry_rel);
    int pn[2] = {endv[0], endv[1]};
    if (act && c + 1 < nch) { const int *p = v3_pred(ax, c + 1) + 2 * pl; pn[0] = p[0]; pn[1] = p[1]; }
    if (!v3_row_equal(endv, pn, dd) || ((ax.debug & 1) && act && c + 1 < nch)) {   // the repaired decoder does not arrive where the unproven one did: the sequential pass goes on from here
      int *f = v3_fix(ax, c + 1) + 2 * pl; f[0] = endv[0]; f[1] = endv[1];
      if (pl == 0) { ax.ctl[V3_CTL_HDR + (ax.cap + 2) + c + 1] = 1; atomicAdd(ax.ctl + V3_CTL_CONFLICT, 1); }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");          // (the next round's staging overwrites this round's LDS)
  }
}
template <int NTB> __global__ __launch_bounds__(64 * V3_WGW) void viterbi_repair_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const RxState *st,
                                                      long long steps_fixed, VitParams vp, V3Aux ax, long long in_base, long long out_lo)
{
  V3_LDS_DECL(V3_WGW)
  const int n = ax.ctl[V3_CTL_MISMATCH];
  const int wv = V3_WGW > 1 ? (int)(threadIdx.x >> 6) : 0;
  const long long rows = (long long)gridDim.x * V3_WGW * 4, row0 = ((long long)blockIdx.x * V3_WGW + wv) * 4;
  if (row0 >= n) return;
  const V3Lds S = {tab_[wv], wbuf_[wv], bests_[wv], lut};
  v3_init_lut(lut, threadIdx.x & 63);
  V3Lane L; v3_init_lane(threadIdx.x & 15, L);
  v3_repair_rows<NTB>(in, out, st ? st->n_vit_steps : steps_fixed, vp, ax, in_base, out_lo, S, L, rows, row0, n);
}

// The sequential pass: ONE decoder walks the flagged chunks in stream order.  From fix[c] it decodes chunk c; where the state it reaches at chunk c + 1 is own[c + 1] the bytes behind
// are the streaming decoder's already, otherwise it goes on through chunk c + 1.  force (a test hook, dvbt_rx_params.viterbi_verify = 3: the parallel pass is not launched): every
// chunk whose own[] is not pred[] is taken, from pred[].  One wavefront, its first row.
template <int NTB> __device__ __forceinline__ void v3_seq_walk(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, long long total_steps, const VitParams &vp, const V3Aux &ax,
                                                               long long in_base, long long out_lo, const V3Lds &S, const V3Lane &L, int force)
{
  const int lane = threadIdx.x & 63, dd = lane >> 4, pl = lane & 15;
  if (lane == 0) { ax.ctl[V3_CTL_ACC] += ax.ctl[V3_CTL_CHUNKS]; ax.ctl[V3_CTL_ACC + 1] += ax.ctl[V3_CTL_MISMATCH]; }   // (the last pass of every repairing launch)
  if (!force && ax.ctl[V3_CTL_CONFLICT] == 0) return;
  const long long total_out = total_steps / 8 - vp.ntb;
  const int B = vp.chunk_bytes;
  long long nch = v3_nchunks(total_out, ax.grid0, B);
  if (nch > ax.cap) nch = ax.cap;
  const int *cflag = ax.ctl + V3_CTL_HDR + (ax.cap + 2);
  int done = 0;
  long long pos = 0;                                                 // everything up to and including chunk pos is the streaming decoder's
  for (;;) {
    // the next chunk behind pos that is flagged (64 candidates per round, one per lane)
    long long cn = -1;
    for (long long c0 = pos + 1; c0 < nch && cn < 0; c0 += 64) {
      const long long c = c0 + lane;
      bool hit = false;
      if (c < nch) hit = cflag[c] != 0 || (force && !v3_slots_equal(v3_own(ax, c), v3_pred(ax, c)));
      const unsigned long long b = __ballot(hit);
      if (b) cn = c0 + __builtin_ctzll(b);
    }
    if (cn < 0) break;
    long long c = cn;
    int v[2];
    { const int *p = (cflag[c] ? v3_fix(ax, c) : v3_pred(ax, c)) + 2 * pl; v[0] = p[0]; v[1] = p[1]; }
    for (;;) {
      const long long b0 = ax.grid0 + c * B;
      const bool act = dd == 0;
      int st0[2] = {v[0], v[1]}, endv[2] = {0, 0};
      if (act) { int *p = v3_pred(ax, c) + 2 * pl; p[0] = v[0]; p[1] = v[1]; }
      int *carry = (act && ax.carry_out && b0 + B >= total_out) ? ax.carry_out + 2 * pl : nullptr;
      v3_decode<NTB, 0, 2>(in, out, total_steps, total_out, vp, in_base, out_lo, b0, act, S, L, st0, endv, v3_own(ax, c) + 2 * pl, nullptr, nullptr, carry, ax.carry_rel);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      done++;
      if (c + 1 >= nch) { pos = nch; break; }
      int on[2]; { const int *p = v3_own(ax, c + 1) + 2 * pl; on[0] = p[0]; on[1] = p[1]; }
      const bool eq = (__ballot(on[0] != endv[0] || on[1] != endv[1]) & 0xffffull) == 0;
      if (act) { int *p = v3_pred(ax, c + 1) + 2 * pl; p[0] = endv[0]; p[1] = endv[1]; }
      if (eq) { pos = c + 1; break; }
      c++; v[0] = endv[0]; v[1] = endv[1];
    }
  }
  if (lane == 0) { ax.ctl[V3_CTL_SEQ] = done; ax.ctl[V3_CTL_ACC + 2] += done; }
}
template <int NTB> __global__ __launch_bounds__(64) void viterbi_repair_seq_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const RxState *st,
                                                      long long steps_fixed, VitParams vp, V3Aux ax, long long in_base, long long out_lo, int force)
{
  V3_LDS_DECL(1)
  const V3Lds S = {tab_[0], wbuf_[0], bests_[0], lut};
  v3_init_lut(lut, threadIdx.x & 63);
  V3Lane L; v3_init_lane(threadIdx.x & 15, L);
  v3_seq_walk<NTB>(in, out, st ? st->n_vit_steps : steps_fixed, vp, ax, in_base, out_lo, S, L, force);
}

// Check, parallel repair (16 decoder rows) and sequential pass in ONE launch of one workgroup, for launches of few chunks (V3_FIX_SMALL): the lock periods of a walk, the single
// block's calls, short pieces -- where three launches that mostly find nothing to do cost more than they compute (BASELINE config 5 at the prescribed noise: 121 Viterbi launches
// per 16 superframes).  The same passes in the same order; the workgroup's barriers stand where the launch boundaries stood.
constexpr long long V3_FIX_SMALL = 1024;
template <int NTB> __global__ __launch_bounds__(64 * V3_WGW) void viterbi_fix_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const RxState *st,
                                                      long long steps_fixed, VitParams vp, V3Aux ax, long long in_base, long long out_lo, int force)
{
  V3_LDS_DECL(V3_WGW)
  const int wv = V3_WGW > 1 ? (int)(threadIdx.x >> 6) : 0;
  const long long total_steps = st ? st->n_vit_steps : steps_fixed;
  const long long nch = v3_nchunks(total_steps / 8 - vp.ntb, ax.grid0, vp.chunk_bytes);
  if (threadIdx.x == 0) ax.ctl[V3_CTL_CHUNKS] = (int)(nch < INT_MAX ? nch : INT_MAX);
  for (long long c = 1 + threadIdx.x; c < nch && c <= ax.cap; c += 64 * V3_WGW) {
    ax.ctl[V3_CTL_HDR + (ax.cap + 2) + c] = 0;
    if (!v3_slots_equal(v3_own(ax, c), v3_pred(ax, c))) ax.ctl[V3_CTL_HDR + atomicAdd(ax.ctl + V3_CTL_MISMATCH, 1)] = (int)c;
  }
  __threadfence(); __syncthreads();
  const int n = *(volatile int *)(ax.ctl + V3_CTL_MISMATCH);
  const V3Lds S = {tab_[wv], wbuf_[wv], bests_[wv], lut};
  v3_init_lut(lut, threadIdx.x & 63);
  V3Lane L; v3_init_lane(threadIdx.x & 15, L);
  if (!force && n > 0) v3_repair_rows<NTB>(in, out, total_steps, vp, ax, in_base, out_lo, S, L, 4 * V3_WGW, 4 * wv, n);
  __threadfence(); __syncthreads();
  if (wv == 0) v3_seq_walk<NTB>(in, out, total_steps, vp, ax, in_base, out_lo, S, L, force);
}

}  // namespace dvbt
