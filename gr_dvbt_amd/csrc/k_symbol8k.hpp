// k_symbol8k.hpp -- the per-OFDM-symbol kernel of the segment path for the 8k mode (A1 tail + A2 + A3 + A4), the
// configuration the headline metric is quoted on (k_symbol2k.hpp is the 2k mode's).  One kernel takes a symbol from its baseband samples to
// its 6048 one-byte labels; the symbol never leaves the CU in between.  Against the demodulator blocks of the block API (demod_kernel,
// k_frontend.hpp) two things differ, both inside the float tolerance of the equalised-carrier tap: the common phasor of frequency_correction
// (:793-819) is not applied (it multiplies pilots and payload alike and cancels in x[c] * ref / x[pilot], and with it the only use of the NEXT
// symbol disappears); the LS gain of an estimation carrier is computed once instead of once per carrier that uses it, and the interpolation
// step (g[R]-g[L])/11 (:625) is a multiplication by 1/11.  How the symbol moves through the CU:
//  * PERSISTENT workgroups (two per CU): a workgroup takes symbol after symbol and requests most of its next symbol's samples (10 of a thread's 16,
//    through a buffer resource: no address registers) while it equalises the current one; the other six, for which there are no registers
//    next to the equaliser, at the top of the next iteration, where they are used last;
//  * symbols are handed out by an atomic counter, so a workgroup that starts late (other kernels share the machine when segments are in flight)
//    takes fewer; the scattered-pilot pattern of a symbol follows from the last one the workgroup saw, and the equaliser's per-carrier table rows
//    (carrier, bracketing pilots, distance: one packed word, L2 resident) are requested for the expected pattern before the pattern search
//    has run and again only when the pattern found differs (lock transients);
//  * the FFT is 8192 = 16 x 16 x (2 x 16) with the first radix-16 pass done on the registers the samples were loaded into
//    (element n = tid + 512 i IS the butterfly of thread tid), and the last pass written in natural, fft-shifted order: no
//    separate digit-reversal pass, 3 LDS stores + 2 LDS loads per point instead of 6 + 5, and 7 barriers per symbol instead of 13;
//  * no padding: XOR swizzles keep every access pattern on distinct banks (see swz1 / swz2) in exactly 64 KB;
//  * derotation phasors from per-symbol tables (2 x 16 + 2 x 32 entries per piece) written one phase ahead by 128 threads: one
//    complex product per thread and piece instead of two sincos;
//  * the integer-CFO search and the pattern search (for all 16 candidate offsets) run side by side on the two halves of the workgroup,
//    every thread then takes the two arg-max itself: one barrier instead of four;
//  * the kernel is bound by VALU issue (every wave64 instruction of a mixed stream costs ~4 cycles, profiles/r02_ubench_mix.json), so the
//    arithmetic is written for instruction COUNT: complex products as v_pk_mul_f32 + v_pk_fma_f32 with op_sel / neg modifiers (2 instructions; the
//    compiler's best is 4 and a third of its FFT passes were register moves), multiplications by the fixed 16th / 32nd roots of unity as
//    modifier variants of the same pair over four (cos, sin) constants, +-i folded into the butterflies' adds;
//  * the demapper first tries the cell of the square grid the carrier falls into: when the carrier is farther than 2e-5 of a cell from every
//    decision boundary (and within 4 cells of the grid) the cell's point is the reference's first strict minimum whatever the rounding of its
//    float distances (bound: 2.5e-6 of a cell, see s8_demap_cell); the 4-candidate search with the reference's tie rule (demap_fast) and the
//    exhaustive one (demap_all) remain for the rest, wave by wave.
// Reference: ofdm_sym_acquisition_impl.cc:285-309,527-534 (derotation), fft_vcc forward + shift (SURVEY C-2),
// reference_signals_impl.cc:536-689,715-744,1065-1124 (pilot engine), dvbt_demap_impl.cc:167-203.
#pragma once
#include "k_frontend.hpp"
#include "k_backend.hpp"

namespace dvbt {

#ifndef S8_EXP
#define S8_EXP 0      // experiment builds only (tools/s8_attribution.sh; wrong output): 1 no pilot engine / demapper, 2 no passes 2 and 3, 4 the next symbol's samples are
                      // loaded at the top of the loop (no prefetch), 8 no equaliser + demapper, 16 no integer-CFO / pattern search, 32 no derotation, 64 the next symbol's samples are requested after the demapper,
                      // 128 only HALF of the symbol's LDS image is allocated (the small arrays in front of it; accesses to the other half fall outside the workgroup's allocation: reads
                      // return 0, stores are dropped) and the demapper's slow path is off: what a kernel that needed 32 KB of LDS for its image would cost AT BEST -- with
                      // -DS8_WG_PER_CU=3 -DS8_MIN_WAVES=6 the occupancy experiment of tools/s8_occupancy.sh
#endif
#ifndef S8_MIN_WAVES
#define S8_MIN_WAVES 4  // __launch_bounds__' wavefronts per SIMD: 4 = 128 VGPRs = two 512-thread workgroups per CU own every register of its SIMDs; 5 -> 96, 6 -> 80 (experiment builds)
#endif
constexpr int S8_N = 8192, S8_T = 512, S8_PAY = 6048, S8_NCP = 177, S8_NTPS = 68, S8_ZL = 688;
constexpr int S8_IT = (S8_PAY + S8_T - 1) / S8_T;             // payload carriers per thread (12)
#ifndef S8_TOP_N
#define S8_TOP_N 6
#endif
constexpr int S8_TOP = S8_TOP_N;                              // of a thread's 16 samples the first S8_TOP are requested at the top of the symbol's iteration (and used last), the others one symbol ahead
constexpr size_t S8_SMALL_BYTES = DEMOD_NP * 8 + 2 * 128 * 8 + 64 * 8 + 192 * 4 + 16 * 4 + 64 * 4 + 192 * 2 + 64 + 16;   // everything but the symbol's image
constexpr size_t S8_LDS_BYTES = (size_t)((S8_EXP & 128) ? S8_N / 2 : S8_N) * 8 + S8_SMALL_BYTES;
#ifndef S8_WG_PER_CU
#define S8_WG_PER_CU 2
#endif
inline int s8_grid(int cus) { return S8_WG_PER_CU * cus; }          // two workgroups per CU (what the LDS holds)

// layout of the first two passes: a = k1 * 512 + (index inside the 512-point sub-transform k1).  Bits 3:0 are XORed with bits 8:5 and bit 4
// with bit 9, so that 16 rows of 32 (stride 32) and two neighbouring sub-transforms land on 32 distinct bank pairs
__device__ __forceinline__ int s8_swz1(int a) { return a ^ ((a >> 5) & 15) ^ (((a >> 9) & 1) << 4); }
// natural (fft-shifted) layout of the spectrum: bits 3:0 XORed with bits 7:4 (stride-16 stores of the last pass, consecutive-carrier reads)
__device__ __forceinline__ int s8_swz2(int b) { return b ^ ((b >> 4) & 15); }

// phasor tables of one symbol (128 entries): [0,16) S_A(i) = expj(512 i incA), [16,32) S_B, [32,48) expj(thA + 32 a incA), [48,64) the same for B,
// [64,96) expj(b incA), [96,128) for B; sample n = tid + 512 i of piece X has phase th_X + n inc_X: the phase is piecewise linear in n (increment incA up to the
// switch position sw, incB after it, ofdm_sym_acquisition_impl.cc:285-309)
__device__ __forceinline__ void s8_fill_ptab(float2 *pt, const SymMeta &m, int t)
{
  const double thA = (double)m.ph_base + m.incA, thB = (double)m.ph_base + (double)m.sw * (m.incA - m.incB) + m.incB;
  double ang;
  if (t < 16) ang = 512.0 * t * m.incA;
  else if (t < 32) ang = 512.0 * (t - 16) * m.incB;
  else if (t < 48) ang = thA + 32.0 * (t - 32) * m.incA;
  else if (t < 64) ang = thB + 32.0 * (t - 48) * m.incB;
  else if (t < 96) ang = (double)(t - 64) * m.incA;
  else ang = (double)(t - 96) * m.incB;
  float sn, cs; sincosf(wrap_pi(ang), &sn, &cs);
  pt[t] = make_float2(cs, sn);
}

// ---- packed complex arithmetic: a float2 lives in an aligned VGPR pair, one VOP3P instruction works on both halves
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f s8_v(float2 a) { return (v2f){a.x, a.y}; }
__device__ __forceinline__ float2 s8_f(v2f a) { return make_float2(a.x, a.y); }
// a * b (2 instructions; the second one is fused: the rounding differs from cmul() in the last bit)
__device__ __forceinline__ v2f s8_cmul(v2f a, v2f b)
{
  v2f p, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(p) : "v"(a), "v"(b));                                        // (a.y b.y, a.y b.x)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(p));          // (a.x b.x - p.x, a.x b.y + p.y)
  return r;
}
__device__ __forceinline__ v2f s8_add_mi(v2f a, v2f b)    // a - i b
{ v2f r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2f s8_add_pi(v2f a, v2f b)    // a + i b
{ v2f r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2f s8_mul_mi(v2f y)           // -i y = (y.y, -y.x)
{ v2f r; asm("v_pk_add_f32 %0, %1, 0 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(y)); return r; }
// y * w for w built from K = (c, s) = (cos t, sin t): the five sign / swap patterns the 16th and 32nd roots need
#define S8_MULK(name, m1, m2) \
  __device__ __forceinline__ v2f name(v2f y, v2f K) { v2f p, r; \
    asm("v_pk_mul_f32 %0, %1, %2 " m1 : "=v"(p) : "v"(y), "v"(K)); \
    asm("v_pk_fma_f32 %0, %1, %2, %3 " m2 : "=v"(r) : "v"(y), "v"(K), "v"(p)); return r; }
S8_MULK(s8_mul_c_ms, "op_sel:[0,0] op_sel_hi:[1,0]", "op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]")                                   // w = ( c, -s)
S8_MULK(s8_mul_s_mc, "op_sel:[0,1] op_sel_hi:[1,1]", "op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]")                                   // w = ( s, -c)
S8_MULK(s8_mul_mc_s, "op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]", "op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]")         // w = (-c,  s)
S8_MULK(s8_mul_ms_mc, "op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]", "op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]")        // w = (-s, -c)
S8_MULK(s8_mul_mc_ms, "op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]", "op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]")        // w = (-c, -s)
#undef S8_MULK
__device__ __forceinline__ v2f s8_mul_h_mh(v2f y, v2f H)  // y * h (1 - i), H = (h, h)
{
  v2f t, r;
  asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(t) : "v"(y));                                    // (y.x + y.y, y.y - y.x)
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(t), "v"(H));
  return r;
}
__device__ __forceinline__ v2f s8_mul_mh_mh(v2f y, v2f H) // y * -h (1 + i)
{
  v2f t, r;
  asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(y));                                    // (y.x - y.y, y.y + y.x)
  asm("v_pk_mul_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(t), "v"(H));
  return r;
}
// the four constants of the fixed twiddles, in VGPR pairs
struct S8Roots { v2f K8, K16, K316, H; };   // (cos, sin) of pi/8, pi/16, 3 pi/16; (sqrt 1/2, sqrt 1/2)
__device__ __forceinline__ S8Roots s8_roots()
{
  S8Roots R;
  R.K8 = (v2f){0.92387953251128674f, 0.38268343236508977f}; R.K16 = (v2f){0.98078528040323043f, 0.19509032201612825f};
  R.K316 = (v2f){0.83146961230254524f, 0.55557023301960218f}; R.H = (v2f){0.70710678118654752f, 0.70710678118654752f};
  return R;
}
__device__ __forceinline__ void s8_bfly4(v2f a0, v2f a1, v2f a2, v2f a3, v2f &y0, v2f &y1, v2f &y2, v2f &y3)
{
  const v2f s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
  y0 = s02 + s13; y2 = s02 - s13; y1 = s8_add_mi(d02, d13); y3 = s8_add_pi(d02, d13);
}
// 16-point forward DFT in registers, as dft16() (k_frontend.hpp): 64 adds + 17 instructions of fixed twiddles
__device__ __forceinline__ void s8_dft16(v2f (&a)[16], const S8Roots &R)
{
  v2f t[16];
#pragma unroll
  for (int rp = 0; rp < 4; rp++) {
    v2f y0, y1, y2, y3;
    s8_bfly4(a[rp], a[rp + 4], a[rp + 8], a[rp + 12], y0, y1, y2, y3);
    t[rp] = y0;
    if (rp == 0) { t[4] = y1; t[8] = y2; t[12] = y3; }
    else if (rp == 1) { t[5] = s8_mul_c_ms(y1, R.K8); t[9] = s8_mul_h_mh(y2, R.H); t[13] = s8_mul_s_mc(y3, R.K8); }            // W16^1, ^2, ^3
    else if (rp == 2) { t[6] = s8_mul_h_mh(y1, R.H); t[10] = s8_mul_mi(y2); t[14] = s8_mul_mh_mh(y3, R.H); }                   // W16^2, ^4, ^6
    else { t[7] = s8_mul_s_mc(y1, R.K8); t[11] = s8_mul_mh_mh(y2, R.H); t[15] = s8_mul_mc_s(y3, R.K8); }                       // W16^3, ^6, ^9
  }
#pragma unroll
  for (int k1 = 0; k1 < 4; k1++) {
    v2f y0, y1, y2, y3;
    s8_bfly4(t[k1 * 4], t[k1 * 4 + 1], t[k1 * 4 + 2], t[k1 * 4 + 3], y0, y1, y2, y3);
    a[k1] = y0; a[k1 + 4] = y1; a[k1 + 8] = y2; a[k1 + 12] = y3;
  }
}
// a[k] *= w1^k, k = 1..15 (powers by products, at most 4 deep, as twiddle16)
__device__ __forceinline__ void s8_twiddle16(v2f (&a)[16], v2f w1)
{
  v2f w[16];
  w[1] = w1;
  w[2] = s8_cmul(w[1], w[1]); w[3] = s8_cmul(w[2], w[1]); w[4] = s8_cmul(w[2], w[2]); w[5] = s8_cmul(w[4], w[1]); w[6] = s8_cmul(w[3], w[3]);
  w[7] = s8_cmul(w[4], w[3]); w[8] = s8_cmul(w[4], w[4]); w[9] = s8_cmul(w[8], w[1]); w[10] = s8_cmul(w[5], w[5]); w[11] = s8_cmul(w[8], w[3]);
  w[12] = s8_cmul(w[6], w[6]); w[13] = s8_cmul(w[8], w[5]); w[14] = s8_cmul(w[7], w[7]); w[15] = s8_cmul(w[8], w[7]);
#pragma unroll
  for (int k = 1; k < 16; k++) a[k] = s8_cmul(a[k], w[k]);
}
// u * W_32^b for the second half (c = 1) of the last pass, b = 1..15
template <int B> __device__ __forceinline__ v2f s8_mul_w32(v2f u, const S8Roots &R)
{
  if (B == 1) return s8_mul_c_ms(u, R.K16); if (B == 2) return s8_mul_c_ms(u, R.K8); if (B == 3) return s8_mul_c_ms(u, R.K316);
  if (B == 4) return s8_mul_h_mh(u, R.H);
  if (B == 5) return s8_mul_s_mc(u, R.K316); if (B == 6) return s8_mul_s_mc(u, R.K8); if (B == 7) return s8_mul_s_mc(u, R.K16);
  if (B == 8) return s8_mul_mi(u);
  if (B == 9) return s8_mul_ms_mc(u, R.K16); if (B == 10) return s8_mul_ms_mc(u, R.K8); if (B == 11) return s8_mul_ms_mc(u, R.K316);
  if (B == 12) return s8_mul_mh_mh(u, R.H);
  if (B == 13) return s8_mul_mc_ms(u, R.K316); if (B == 14) return s8_mul_mc_ms(u, R.K8); if (B == 15) return s8_mul_mc_ms(u, R.K16);
  return u;
}

// sum / maximum over the 16 lanes of a DPP row (every lane gets it), over a quad
template <int CTRL> __device__ __forceinline__ float s8_dpp(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true)); }
__device__ __forceinline__ float s8_row_sum(float v)
{ v += s8_dpp<0xB1>(v); v += s8_dpp<0x4E>(v); v += s8_dpp<0x141>(v); v += s8_dpp<0x140>(v); return v; }   // quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float s8_row_max(float v)
{ v = fmaxf(v, s8_dpp<0xB1>(v)); v = fmaxf(v, s8_dpp<0x4E>(v)); v = fmaxf(v, s8_dpp<0x141>(v)); v = fmaxf(v, s8_dpp<0x140>(v)); return v; }
__device__ __forceinline__ float s8_quad_max(float v) { v = fmaxf(v, s8_dpp<0xB1>(v)); return fmaxf(v, s8_dpp<0x4E>(v)); }

// A4, first try: the cell of the nlev x nlev grid the carrier falls into.  g = e / step + nlev / 2 puts level j on [j, j + 1); outside the grid the
// edge cells extend outwards.  The cell's point is the reference's answer (first strict minimum of fl(fl(dx^2) + fl(dy^2)) over all points,
// dvbt_demap_impl.cc:167-203) whenever the carrier is farther from the cell's boundaries than rounding can reorder two neighbours: t cells inside
// the nearer boundary the neighbouring level's squared distance is larger by 2 t step^2; against that stand the rounding of dx (2.4e-7 step^2 after
// the square, |x| <= 8 step), of the squares (1.5e-8 step^2) and of the two sums (6e-8 (ax + ay) each, ay <= 4.5^2 step^2 inside S8_REACH):
// together < 3e-6 step^2, so t > 1.5e-6 decides; the g computed here is off by <= 1e-6.  S8_MARGIN = 2e-5 (eight times that).
// Returns false when the carrier is closer than S8_MARGIN to a boundary or farther than S8_REACH cells from the grid: demap_fast / demap_all decide.
constexpr float S8_MARGIN = 2.0e-5f, S8_REACH = 4.0f;
__device__ __forceinline__ bool s8_demap_cell(v2f e, float inv_step, float half_n, float top, int &idx)
{
  const float gx = __builtin_fmaf(e.x, inv_step, half_n), gy = __builtin_fmaf(e.y, inv_step, half_n);
  const float cx = __builtin_amdgcn_fmed3f(gx, 0.5f, top), cy = __builtin_amdgcn_fmed3f(gy, 0.5f, top);     // top = nlev - 0.5
  const float fx = floorf(cx), fy = floorf(cy);
  const bool ok = fabsf((cx - fx) - 0.5f) < 0.5f - S8_MARGIN && fabsf((cy - fy) - 0.5f) < 0.5f - S8_MARGIN &&
                  fabsf(gx - cx) < S8_REACH && fabsf(gy - cy) < S8_REACH;
  idx = (int)fx * 8 + (int)fy;
  return ok;
}

// The 16 samples of a thread are read through a buffer resource whose base is the symbol's first sample (scalar registers): every load is
// descriptor + the thread's 32-bit offset + a scalar offset i * 4096, no 64-bit address registers per load (those were what spilled next to the
// equaliser, each spill behind an s_waitcnt that stalled the wavefront for a memory round trip)
typedef int s8_i4 __attribute__((ext_vector_type(4)));
__device__ v2f s8_raw_buffer_load_v2f32(s8_i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ __forceinline__ s8_i4 s8_rsrc(const v2f *base)
{
  const unsigned long long a = (unsigned long long)base;
  s8_i4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a); r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) & 0xffff;   // 48-bit base, stride 0
  r.z = 0x7fffffff; r.w = 0x00020000;                                                                                                  // no range limit that matters; gfx9 raw 32-bit format
  return r;
}
// the same with the buffer's range set to what is really in memory behind `low` (FrontParams.avail): a symbol whose window has crept beyond the segment's
// end (k_frontend.hpp) reads zeros there -- the hardware's range check, no instruction in the load path
__device__ __forceinline__ s8_i4 s8_rsrc_lim(const v2f *iq, long long low, long long avail)
{
  s8_i4 r = s8_rsrc(iq + low);
  long long bytes = (avail - low) * 8; bytes = bytes < 0 ? 0 : (bytes > 0x7fffffffll ? 0x7fffffffll : bytes);
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  return r;
}
__device__ __forceinline__ v2f s8_sample(s8_i4 rsrc, int i, int tid) { return s8_raw_buffer_load_v2f32(rsrc, tid * 8, i * S8_T * 8, 0); }

// TAPS: the debug taps (derotated samples, spectrum, equalised carriers) are compiled in; the production instantiation has none of that code
// DRIFT: the instantiation that also reproduces the wander of the reference's float phase accumulator (k_drift.hpp): the derotation phasor of a sample is
// multiplied by (1 + i delta) of its 32-sample block.  Both instantiations are launched; the device-side flag drift_flags[1] decides which one works
// (the other returns before it takes a symbol), so the path without a carrier offset keeps its registers and instructions.
template <bool TAPS, bool DRIFT> __global__ __launch_bounds__(S8_T, S8_MIN_WAVES) void symbol8k_kernel(const float2 *__restrict__ iq_, FrontParams p, const RxState *st,
                                                           const SymMeta *__restrict__ meta, const float2 *__restrict__ tw, float2 *__restrict__ acq_tap,
                                                           float2 *__restrict__ fft_tap, DemodTables T, float2 *__restrict__ eq_tap,
                                                           float2 *__restrict__ tpsval, SymInfo *__restrict__ info, InnerParams ip,
                                                           const float2 *__restrict__ points, const unsigned char *__restrict__ label_tab,
                                                           uint8_t *__restrict__ labels, int *__restrict__ ticket,
                                                           const float *__restrict__ drift, const int *__restrict__ drift_flags, float *__restrict__ csi_tap)
{
  if ((drift_flags[1] != 0) != DRIFT) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  v2f *x = reinterpret_cast<v2f *>((S8_EXP & 128) ? smem_raw + S8_SMALL_BYTES : smem_raw);
  v2f *gtab = (S8_EXP & 128) ? reinterpret_cast<v2f *>(smem_raw) : x + S8_N;   // LS gains at the estimation carriers
  float2 *ptab = reinterpret_cast<float2 *>(gtab + DEMOD_NP);    // [2][128] phasor tables, this symbol's and the next one's
  float2 *pts = ptab + 256;
  float *s_known = reinterpret_cast<float *>(pts + 64);          // 192
  float *s_cfo = s_known + 192;                                  // 16 candidate offsets
  float *s_pat = s_cfo + 16;                                     // [16 candidates][4 patterns]
  short *s_cpil = reinterpret_cast<short *>(s_pat + 64);         // 192
  unsigned char *label_of = reinterpret_cast<unsigned char *>(s_cpil + 192);
  int *s_tkt = reinterpret_cast<int *>(label_of + 64);          // the symbol this workgroup takes next
  const v2f *iq = reinterpret_cast<const v2f *>(iq_);
  constexpr int N = S8_N, zl = S8_ZL;
  const int tid0 = threadIdx.x, tid = tid0, cp = p.cp;
  const int nsym = st->n_symbols, call0 = st->call0;             // read once: st may alias the stores below, every re-read would be a vector load and a full wait
  // symbols are handed out by a counter (zeroed by the host before the launch), not by a fixed stride: with other kernels sharing the machine the
  // workgroups of this launch start at different times, and one that starts late simply takes fewer symbols
  if (tid == 0) s_tkt[0] = atomicAdd(ticket, 1);
  __syncthreads();
  int s = __builtin_amdgcn_readfirstlane(s_tkt[0]);
  if (s >= nsym) return;

  if (tid < 64) { pts[tid] = points[tid]; label_of[tid] = label_tab[tid]; }
  if (tid < S8_NCP) { s_cpil[tid] = T.cpilot[tid]; if (tid < S8_NCP - 1) s_known[tid] = T.known_diff[tid]; }
  const v2f w1A = s8_v(tw[tid]), w1B = s8_v(tw[16 * (tid & 31)]); // W_8192^n2 (first pass), W_512^m2 (second pass): the thread's twiddle bases
  const int tps_c = tid < S8_NTPS ? T.tps[tid] : 0;
  float pat_ref = 0.f;                                            // reference value of this thread's scattered pilot in the pattern search
  if (tid >= 256) { const int q = tid - 256, j = q & 15, pat = (q >> 4) & 3; if (j < 10) pat_ref = T.pilot_ref[3 * pat + 12 * j]; }
  const float half_n = 0.5f * (float)ip.nlev, top = (float)ip.nlev - 0.5f;

  // the pattern (symbol index mod 4) of a symbol of a locked stream follows from the last one's: the per-pattern table rows of a symbol (T.pay_pack
  // and the estimation-carrier list, L2 resident) are requested before the pattern search has confirmed it
  int cur_mod = 0, s_prev = s;

  SymMeta m = meta[s];
  v2f vin[16];
  {
    const long long low = (long long)(call0 + s) * (N + cp) + m.cp_start - N + 1;
    const s8_i4 rs = s8_rsrc_lim(iq, low, p.avail);
#pragma unroll
    for (int i = S8_TOP; i < 16; i++) vin[i] = s8_sample(rs, i, tid);
  }
  if (tid < 128) s8_fill_ptab(ptab, m, tid);
  int par = 0;
  __syncthreads();

  for (;;) {
    // the thread's index and twiddle bases, opaque to the optimiser: everything derived from them is a handful of instructions, and hoisting it out of
    // the loop (the 30 twiddle powers, ~80 LDS addresses, the root constants) costs more registers than the loop has
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const S8Roots R = s8_roots();
    int tkt = 0;
    if (tid == 0) tkt = atomicAdd(ticket, 1);                     // the next symbol of this workgroup; published through LDS at the end of the FFT
    const bool last = !p.keep_last && s + 1 >= nsym;             // no output for the last item (the reference's demod consumes n+1 items)
    {   // the samples that did not fit into the registers next to the equaliser (the compiler spilled them, each behind a full wait): they are
        // requested now and used last in the derotation below
      const long long low = (long long)(call0 + s) * (N + cp) + m.cp_start - N + 1;
      const s8_i4 rs = s8_rsrc_lim(iq, low, p.avail);
#pragma unroll
      for (int i = 0; i < ((S8_EXP & 4) ? 16 : S8_TOP); i++) vin[i] = s8_sample(rs, i, tid);
    }
    float dl[16];
    if (DRIFT) {
      const float *dt = drift + (size_t)s * (S8_N / 32) + (tid >> 5);           // sample n = tid + 512 i lies in block (tid >> 5) + 16 i
#pragma unroll
      for (int i = 0; i < 16; i++) dl[i] = dt[16 * i];
    }
    // ---- A1 tail: derotate (ofdm_sym_acquisition_impl.cc:285-309,527-534), on the registers the loads arrived in
    v2f a[16];
    {
      const v2f *pt = reinterpret_cast<const v2f *>(ptab) + par * 128;
      const v2f PA = s8_cmul(pt[32 + (tid >> 5)], pt[64 + (tid & 31)]), PB = s8_cmul(pt[48 + (tid >> 5)], pt[96 + (tid & 31)]);
      const int sw = (m.sw >= 0 && m.sw < N + cp) ? m.sw : 0x7fffffff;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int i = (k + S8_TOP) & 15, n = tid + i * S8_T;
        const bool pieceB = n + 1 > sw;
        const v2f P = pieceB ? PB : PA;
        a[i] = (S8_EXP & 32) ? vin[i] : s8_cmul(s8_cmul(P, pt[(pieceB ? 16 : 0) + i]), vin[i]);
        if (DRIFT) a[i] = (v2f){__builtin_fmaf(-dl[i], a[i].y, a[i].x), __builtin_fmaf(dl[i], a[i].x, a[i].y)};    // x (1 + i delta), |delta| < 2e-3
        if (TAPS && acq_tap) acq_tap[(size_t)s * N + n] = s8_f(a[i]);
      }
    }
    // ---- A2, pass 1: n = n2 + 512 n1 -> Y[k1][n2] W_8192^(n2 k1)
    s8_dft16(a, R);
    { v2f wA = w1A; asm volatile("" : "+v"(wA), "+v"(a[15]));   // the power chain starts here, not above the butterfly (its 30 registers do not fit there)
      s8_twiddle16(a, wA); }
    __syncthreads();                                             // the previous symbol's readers of x are done
    {
      const int b0 = tid ^ ((tid >> 5) & 15);
#pragma unroll
      for (int k1 = 0; k1 < 16; k1++) x[k1 * 512 + (b0 ^ ((k1 & 1) << 4))] = a[k1];
    }
    __syncthreads();
    if (!(S8_EXP & 2)) {
    // ---- pass 2: the 512-point transform of row k1, n2 = 32 m1 + m2 -> Z[k1][j1][m2] W_512^(m2 j1), in place
    {
      const int k1 = tid >> 5, mm = (tid & 31) ^ ((k1 & 1) << 4), rb = k1 * 512;
#pragma unroll
      for (int m1 = 0; m1 < 16; m1++) a[m1] = x[rb + 32 * m1 + (mm ^ m1)];
      s8_dft16(a, R);
      { v2f wB = w1B; asm volatile("" : "+v"(wB), "+v"(a[15]));
        s8_twiddle16(a, wB); }
#pragma unroll
      for (int j1 = 0; j1 < 16; j1++) x[rb + 32 * j1 + (mm ^ j1)] = a[j1];
    }
    __syncthreads();
    // ---- pass 3: the 32-point transform of row (k1, j1): radix 2 (wave-uniform half c) then 16; bin k1 + 16 j1 + 256 c + 512 d
    {
      const int j1 = tid & 15, k1 = (tid >> 4) & 15, c = tid >> 8, jj = j1 ^ ((k1 & 1) << 4), rb = k1 * 512 + 32 * j1;
      if (c == 0) {
#pragma unroll
        for (int b = 0; b < 16; b++) a[b] = x[rb + (b ^ jj)] + x[rb + ((b + 16) ^ jj)];
      } else {
        v2f u[16];
#pragma unroll
        for (int b = 0; b < 16; b++) u[b] = x[rb + (b ^ jj)] - x[rb + ((b + 16) ^ jj)];
        a[0] = u[0];
        a[1] = s8_mul_w32<1>(u[1], R); a[2] = s8_mul_w32<2>(u[2], R); a[3] = s8_mul_w32<3>(u[3], R); a[4] = s8_mul_w32<4>(u[4], R);
        a[5] = s8_mul_w32<5>(u[5], R); a[6] = s8_mul_w32<6>(u[6], R); a[7] = s8_mul_w32<7>(u[7], R); a[8] = s8_mul_w32<8>(u[8], R);
        a[9] = s8_mul_w32<9>(u[9], R); a[10] = s8_mul_w32<10>(u[10], R); a[11] = s8_mul_w32<11>(u[11], R); a[12] = s8_mul_w32<12>(u[12], R);
        a[13] = s8_mul_w32<13>(u[13], R); a[14] = s8_mul_w32<14>(u[14], R); a[15] = s8_mul_w32<15>(u[15], R);
      }
      s8_dft16(a, R);
      __syncthreads();                                           // every row has been read
      const int base2 = (k1 ^ j1) + 16 * j1 + 256 * c;           // s8_swz2 of the shifted bin: out[b] = X[(b - N/2) mod N]
#pragma unroll
      for (int d = 0; d < 16; d++) x[base2 + 512 * (d ^ 8)] = a[d];
    }
    }
    if (tid == 0) s_tkt[1] = tkt;
    __syncthreads();
    const int s_next = __builtin_amdgcn_readfirstlane(s_tkt[1]);
    const bool more = s_next < nsym;
    SymMeta mn = m;
    if (more) mn = meta[s_next];
    auto X = [&](int b) -> v2f { return x[s8_swz2(b)]; };
    unsigned tcl[S8_IT];                                          // carrier | rank of the left bracketing estimation carrier << 13 | distance to it << 23
    unsigned est0 = 0, est1 = 0, tps_l = 0, tps_dd = 0; int np = 0; // this thread's two estimation carriers, its TPS carrier's left bracket and distance, the pattern's estimation-carrier count
                                                                  // (kept apart: packing them would need the loaded values at once, i.e. a full wait right behind the requests)
    auto load_rows = [&](int md) {                                // the table rows of pattern md
      const uint32_t *pp = T.pay_pack + (size_t)md * S8_PAY;
#pragma unroll
      for (int it = 0; it < S8_IT; it++) { const int i = tid + it * S8_T; tcl[it] = pp[i < S8_PAY ? i : S8_PAY - 1]; }
      np = md == 0 ? T.np[0] : md == 1 ? T.np[1] : md == 2 ? T.np[2] : T.np[3];
      const uint16_t *pk = T.pil_k + (size_t)md * DEMOD_NP;       // carrier | sign of its reference << 15
      est0 = pk[tid < np ? tid : 0]; est1 = pk[tid + S8_T < np ? tid + S8_T : 0];
      if (tid < S8_NTPS) { const int q = md * S8_NTPS + tid; tps_l = T.tps_Li[q]; tps_dd = T.tps_d[q]; }
    };
    const int pred = (cur_mod + (s - s_prev)) & 3;                // what a locked stream will show
    if (!last && !(S8_EXP & 1)) load_rows(pred);
    if (TAPS && fft_tap) {
#pragma unroll
      for (int i = 0; i < 16; i++) { const int b = tid + i * S8_T; fft_tap[(size_t)s * N + b] = s8_f(X(b)); }
    }
    if (last) break;
    if (S8_EXP & 1) {
      if (more) {
        const long long low = (long long)(call0 + s_next) * (N + cp) + mn.cp_start - N + 1;
        const s8_i4 rs = s8_rsrc_lim(iq, low, p.avail);
#pragma unroll
        for (int i = S8_TOP; i < 16; i++) vin[i] = s8_sample(rs, i, tid);
      }
      if (x[tid].x == 123.f) labels[s] = 1;
      if (!more) break;
      s = s_next; m = mn; par ^= 1;
      continue;
    }

    // ---- A3: integer CFO (process_cpilot_data :715-744, 16 candidate shifts x 176 pilot pairs) on threads 0..255, and the symbol index mod 4
    // (process_spilot_data :549-582, first 10 scattered pilots of each pattern) for every candidate shift on threads 256..511
    if (S8_EXP & 16) { if (tid < 80) s_cfo[tid] = tid == 8 ? 1.f : 0.f; }
    else if (tid < 256) {
      const int cand = tid >> 4, sub = tid & 15, i = zl - 8 + cand;
      constexpr int NJ = (S8_NCP - 1) / 16;                         // 176 pairs = 11 per lane: all reads in flight together
      int c0[NJ], c1[NJ]; v2f u[NJ], v[NJ];
#pragma unroll
      for (int k = 0; k < NJ; k++) { c0[k] = s_cpil[sub + 16 * k]; c1[k] = s_cpil[sub + 16 * k + 1]; }
#pragma unroll
      for (int k = 0; k < NJ; k++) { u[k] = X(i + c1[k]); v[k] = X(i + c0[k]); }
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < NJ; k++) { const v2f d = u[k] - v[k]; sum += s_known[sub + 16 * k] * (d.x * d.x + d.y * d.y); }
      sum = s8_row_sum(sum);
      if (sub == 0) s_cfo[cand] = sum;
    } else {
      const int q = tid - 256, j = q & 15, pat = (q >> 4) & 3, cg = q >> 6;
      v2f v[4];
#pragma unroll
      for (int cc = 0; cc < 4; cc++) v[cc] = X(zl - 8 + cg * 4 + cc + 3 * pat + 12 * (j < 10 ? j : 0));
#pragma unroll
      for (int cc = 0; cc < 4; cc++) {
        const float cr = s8_row_sum(j < 10 ? pat_ref * v[cc].x : 0.f), ci = s8_row_sum(j < 10 ? -pat_ref * v[cc].y : 0.f);   // ref * conj(v)
        if (j == 0) s_pat[(cg * 4 + cc) * 4 + pat] = cr * cr + ci * ci;
      }
      // the next symbol's phasor tables, by two wavefronts of the half with the lighter search
      if (more && q < 128) s8_fill_ptab(ptab + (par ^ 1) * 128, mn, q);
    }
    __syncthreads();
    // the two arg-max (first index of the maximum, if positive: the reference's strict > from 0), by every wavefront for itself
    int fo, mod;
    {
      const int l = tid & 63;
      float v = s_cfo[l & 15];
      float mx = s8_row_max(v);
      const unsigned long long hit = __ballot(v == mx && mx > 0.f && l < 16);
      const int best = hit ? __builtin_ctzll(hit) : 8;
      v = s_pat[best * 4 + (l & 3)];
      mx = s8_quad_max(v);
      const unsigned long long hit2 = __ballot(v == mx && mx > 0.f && l < 4);
      fo = best - 8; mod = hit2 ? __builtin_ctzll(hit2) : 0;
    }
    const int xb = zl + fo;
    if (tid == 0) { SymInfo si; si.freq_offset = fo; si.mod_index = mod; si.cfc = 0.f; si.pad = 0; info[s] = si; }
    // the next symbol's samples start travelling now (all but the S8_TOP that are requested at the top of its iteration)
    if (more && !(S8_EXP & (4 | 64))) {
      const long long low = (long long)(call0 + s_next) * (N + cp) + mn.cp_start - N + 1;
      const s8_i4 rs = s8_rsrc_lim(iq, low, p.avail);
#pragma unroll
      for (int i = S8_TOP; i < 16; i++) vin[i] = s8_sample(rs, i, tid);
    }
    if (mod != pred) load_rows(mod);                              // workgroup-uniform; in lock only at a workgroup's first symbol
    cur_mod = mod; s_prev = s;
    // LS gains at the estimation carriers (set_channel_gain :486-490): ref / X = ref conj(X) / |X|^2, ref = +-4/3
    {
      const float amp = (float)(4.0 / 3.0);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int e = (int)(h ? est1 : est0), r = tid + h * S8_T;
        if (r < np) {
          const v2f v = X(xb + (e & 0x7fff));
          const float q = ((e & 0x8000) ? -amp : amp) * __builtin_amdgcn_rcpf(v.x * v.x + v.y * v.y);
          gtab[r] = (v2f){q * v.x, -q * v.y};
        }
      }
    }
    __syncthreads();
    // interpolation (:617-642, the constant 11 of :625) + equalise (:1111-1114) + demap
    auto equalise = [&](int c, int Li, int dj) -> v2f {
      const v2f gl = gtab[Li], gr = gtab[Li + 1];
      const float k11 = 1.0f / 11.0f, j = (float)dj;
      const v2f t = (gr - gl) * k11;
      return s8_cmul(X(xb + c), (v2f){__builtin_fmaf(t.x, j, gl.x), __builtin_fmaf(t.y, j, gl.y)});
    };
    {
      uint8_t *lab = labels + (size_t)s * S8_PAY;
      bool slow = ip.hshift != 0.f;                              // hierarchical constellations: every carrier through the candidate search on the shifted grid
#pragma unroll
      for (int it = 0; it < S8_IT; it++) {
        const int i = tid + it * S8_T;
        if ((it < S8_PAY / S8_T || i < S8_PAY) && !(S8_EXP & 8)) {
          const v2f e = equalise((int)(tcl[it] & 0x1fffu), (int)((tcl[it] >> 13) & 0x3ffu), (int)(tcl[it] >> 23));
          if (TAPS && eq_tap) eq_tap[(size_t)s * S8_PAY + i] = s8_f(e);
          if (TAPS && csi_tap) {   // channel state of the carrier for the soft-decision path (k_soft.hpp): |H|^2 = 1 / |interpolated equaliser gain|^2
            const int Li = (int)((tcl[it] >> 13) & 0x3ffu); const float j = (float)(tcl[it] >> 23);
            const v2f gl = gtab[Li], gr = gtab[Li + 1], t2 = (gr - gl) * (1.0f / 11.0f);
            const float gx = __builtin_fmaf(t2.x, j, gl.x), gy = __builtin_fmaf(t2.y, j, gl.y);
            csi_tap[(size_t)s * S8_PAY + i] = 1.0f / (gx * gx + gy * gy);
          }
          int idx;
          const bool cell_ok = s8_demap_cell(e, ip.inv_step, half_n, top, idx);
          if (!(S8_EXP & 128)) slow |= !cell_ok;
          lab[i] = label_of[idx & 63];
        }
#ifdef S8_BATCH
        if (it % S8_BATCH == S8_BATCH - 1) __builtin_amdgcn_sched_barrier(0);   // carriers in flight together: bounds the registers of this loop next to the prefetched samples
#endif
      }
      // carriers within rounding of a decision boundary or far outside the grid: the 4-candidate search with the reference's tie rule, then the
      // exhaustive one; the wavefront simply takes its carriers again
      if (__any(slow)) {
#pragma unroll 1
        for (int it = 0; it < S8_IT; it++) {
          const int i = tid + it * S8_T;
          if (i >= S8_PAY) break;
          const unsigned w = T.pay_pack[(size_t)mod * S8_PAY + i];
          const v2f e = equalise((int)(w & 0x1fffu), (int)((w >> 13) & 0x3ffu), (int)(w >> 23));
          int f = demap_fast(s8_f(e), pts, label_of, ip);
          if (f < 0) f = demap_all(s8_f(e), pts, ip.csize);
          lab[i] = (uint8_t)f;
        }
      }
    }
    if (more && (S8_EXP & 64)) {
      const long long low = (long long)(call0 + s_next) * (N + cp) + mn.cp_start - N + 1;
      const s8_i4 rs = s8_rsrc_lim(iq, low, p.avail);
#pragma unroll
      for (int i = S8_TOP; i < 16; i++) vin[i] = s8_sample(rs, i, tid);
    }
    if (tid < S8_NTPS)    // equalised TPS carriers (process_tps_data :929-931)
      tpsval[(size_t)s * S8_NTPS + tid] = s8_f(equalise(tps_c, (int)tps_l, (int)tps_dd));
    if (!more) break;
    s = s_next; m = mn; par ^= 1;
  }
}

}  // namespace dvbt
