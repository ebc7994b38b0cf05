// k_symbol8k.hpp -- the per-OFDM-symbol kernel of the segment path for the 8k mode (A1 tail + A2 + A3 + A4), the
// configuration the headline metric is quoted on.  Same results as derot_fft_demod_kernel (k_symbol.hpp, which stays the kernel of
// the 2k mode); what differs is how the symbol moves through the CU:
//  * PERSISTENT workgroups (two per CU): workgroup w takes symbols w, w + G, w + 2G, ... and loads the samples of its next symbol
//    into registers while it equalises the current one, so the HBM latency of a symbol's 64 KB is never exposed;
//  * with G a multiple of 4 a workgroup meets the same scattered-pilot pattern every time: the equaliser's per-carrier table rows
//    (carrier, bracketing pilots, distance) stay in registers and are reloaded only when the pattern found differs (lock transients);
//  * the FFT is 8192 = 16 x 16 x (2 x 16) with the first radix-16 pass done on the registers the samples were loaded into
//    (element n = tid + 512 i IS the butterfly of thread tid), and the last pass written in natural, fft-shifted order: no
//    separate digit-reversal pass, 3 LDS stores + 2 LDS loads per point instead of 6 + 5, and 7 barriers per symbol instead of 13;
//  * no padding: XOR swizzles keep every access pattern on distinct banks (see swz1 / swz2) in exactly 64 KB;
//  * derotation phasors from per-symbol tables (2 x 16 + 2 x 32 entries per piece) written one phase ahead by 128 threads: one
//    complex product per thread and piece instead of two sincos;
//  * the integer-CFO search and the pattern search (for all 16 candidate offsets) run side by side on the two halves of the workgroup,
//    every thread then takes the two arg-max itself: one barrier instead of four.
// Reference: ofdm_sym_acquisition_impl.cc:285-309,527-534 (derotation), fft_vcc forward + shift (SURVEY C-2),
// reference_signals_impl.cc:536-689,715-744,1065-1124 (pilot engine), dvbt_demap_impl.cc:167-203.
#pragma once
#include "k_symbol.hpp"

namespace dvbt {

#ifndef S8_EXP
#define S8_EXP 0      // experiment builds only (tools/s8_attribution.sh; wrong output): 1 no pilot engine / demapper, 2 no passes 2 and 3, 4 the next symbol's samples are
                      // loaded at the top of the loop (no prefetch), 8 no equaliser + demapper, 16 no integer-CFO / pattern search, 32 no derotation
#endif
constexpr int S8_N = 8192, S8_T = 512, S8_PAY = 6048, S8_NCP = 177, S8_NTPS = 68, S8_ZL = 688;
constexpr int S8_IT = (S8_PAY + S8_T - 1) / S8_T;             // payload carriers per thread (12)
constexpr size_t S8_LDS_BYTES = (size_t)S8_N * 8 + DEMOD_NP * 8 + 2 * 128 * 8 + 64 * 8 + 192 * 4 + 16 * 4 + 64 * 4 + 192 * 2 + 64;
inline int s8_grid(int cus) { return (2 * cus) & ~3; }        // two workgroups per CU, a multiple of the pattern period

// layout of the first two passes: a = k1 * 512 + (index inside the 512-point sub-transform k1).  Bits 3:0 are XORed with bits 8:5 and bit 4
// with bit 9, so that 16 rows of 32 (stride 32) and two neighbouring sub-transforms land on 32 distinct bank pairs
__device__ __forceinline__ int s8_swz1(int a) { return a ^ ((a >> 5) & 15) ^ (((a >> 9) & 1) << 4); }
// natural (fft-shifted) layout of the spectrum: bits 3:0 XORed with bits 7:4 (stride-16 stores of the last pass, consecutive-carrier reads)
__device__ __forceinline__ int s8_swz2(int b) { return b ^ ((b >> 4) & 15); }

// phasor tables of one symbol (128 entries): [0,16) S_A(i) = expj(512 i incA), [16,32) S_B, [32,48) expj(thA + 32 a incA), [48,64) the same for B,
// [64,96) expj(b incA), [96,128) for B; sample n = tid + 512 i of piece X has phase th_X + n inc_X (see k_symbol.hpp)
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

// TAPS: the debug taps (derotated samples, spectrum, equalised carriers) are compiled in; the production instantiation has none of that code
template <bool TAPS> __global__ __launch_bounds__(S8_T, 4) void symbol8k_kernel(const float2 *__restrict__ iq, FrontParams p, const RxState *st,
                                                           const SymMeta *__restrict__ meta, const float2 *__restrict__ tw, float2 *__restrict__ acq_tap,
                                                           float2 *__restrict__ fft_tap, DemodTables T, float2 *__restrict__ eq_tap,
                                                           float2 *__restrict__ tpsval, SymInfo *__restrict__ info, InnerParams ip,
                                                           const float2 *__restrict__ points, const unsigned char *__restrict__ label_tab,
                                                           uint8_t *__restrict__ labels)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2 *x = reinterpret_cast<float2 *>(smem_raw);
  float2 *gtab = x + S8_N;                                       // LS gains at the estimation carriers
  float2 *ptab = gtab + DEMOD_NP;                                // [2][128] phasor tables, this symbol's and the next one's
  float2 *pts = ptab + 256;
  float *s_known = reinterpret_cast<float *>(pts + 64);          // 192
  float *s_cfo = s_known + 192;                                  // 16 candidate offsets
  float *s_pat = s_cfo + 16;                                     // [16 candidates][4 patterns]
  short *s_cpil = reinterpret_cast<short *>(s_pat + 64);         // 192
  unsigned char *label_of = reinterpret_cast<unsigned char *>(s_cpil + 192);
  constexpr int N = S8_N, zl = S8_ZL;
  const int tid0 = threadIdx.x, tid = tid0, G = gridDim.x, cp = p.cp;
  const int nsym = st->n_symbols;
  int s = blockIdx.x;
  if (s >= nsym) return;

  if (tid < 64) { pts[tid] = points[tid]; label_of[tid] = label_tab[tid]; }
  if (tid < S8_NCP) { s_cpil[tid] = T.cpilot[tid]; if (tid < S8_NCP - 1) s_known[tid] = T.known_diff[tid]; }
  const float2 w1A = tw[tid], w1B = tw[16 * (tid & 31)];         // W_8192^n2 (first pass), W_512^m2 (second pass): the thread's twiddle bases
  const int tps_c = tid < S8_NTPS ? T.tps[tid] : 0;
  float pat_ref = 0.f;                                            // reference value of this thread's scattered pilot in the pattern search
  if (tid >= 256) { const int q = tid - 256, j = q & 15, pat = (q >> 4) & 3; if (j < 10) pat_ref = T.pilot_ref[3 * pat + 12 * j]; }

  // per-pattern table rows of this thread, kept across symbols
  int cur_mod = -1;
  unsigned tcl[S8_IT];                                            // carrier | rank of the left bracketing estimation carrier << 16  (the right one is rank + 1)
  unsigned tdp[(S8_IT + 3) / 4];                                  // distance to the left estimation carrier, one byte per carrier
  unsigned est01 = 0, tps_ld = 0; int np = 0;

  SymMeta m = meta[s];
  float2 vin[16];
  {
    const long long low = (long long)(st->call0 + s) * (N + cp) + m.cp_start - N + 1;
#pragma unroll
    for (int i = 0; i < 16; i++) vin[i] = iq[low + tid + i * S8_T];
  }
  if (tid < 128) s8_fill_ptab(ptab, m, tid);
  int par = 0;
  __syncthreads();

  for (;;) {
    // the thread's index and twiddle bases, opaque to the optimiser: everything derived from them is a handful of instructions, and hoisting it out of
    // the loop (the 30 twiddle powers, ~80 LDS addresses) costs more registers than the loop has
    int tid = tid0; float2 wA = w1A, wB = w1B;
    asm volatile("" : "+v"(tid), "+v"(wA.x), "+v"(wA.y), "+v"(wB.x), "+v"(wB.y));
    const int s_next = s + G;
    const bool more = s_next < nsym;
    const bool last = !p.keep_last && s + 1 >= nsym;             // no output for the last item (the reference's demod consumes n+1 items)
    SymMeta mn = m;
    if (more) mn = meta[s_next];
    if ((S8_EXP & 4) && s != (int)blockIdx.x) {
      const long long low = (long long)(st->call0 + s) * (N + cp) + m.cp_start - N + 1;
#pragma unroll
      for (int i = 0; i < 16; i++) vin[i] = iq[low + tid + i * S8_T];
    }
    // ---- A1 tail: derotate (ofdm_sym_acquisition_impl.cc:285-309,527-534), on the registers the loads arrived in
    float2 a[16];
    {
      const float2 *pt = ptab + par * 128;
      const float2 PA = cmul(pt[32 + (tid >> 5)], pt[64 + (tid & 31)]), PB = cmul(pt[48 + (tid >> 5)], pt[96 + (tid & 31)]);
      const bool has_sw = m.sw >= 0 && m.sw < N + cp;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int n = tid + i * S8_T;
        const bool pieceB = has_sw && n + 1 > m.sw;
        a[i] = (S8_EXP & 32) ? vin[i] : cmul(cmul(pieceB ? PB : PA, pt[(pieceB ? 16 : 0) + i]), vin[i]);
        if (TAPS && acq_tap) acq_tap[(size_t)s * N + n] = a[i];
      }
    }
    // ---- A2, pass 1: n = n2 + 512 n1 -> Y[k1][n2] W_8192^(n2 k1)
    dft16(a);
    twiddle16(a, wA);
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
      dft16(a);
      twiddle16(a, wB);
#pragma unroll
      for (int j1 = 0; j1 < 16; j1++) x[rb + 32 * j1 + (mm ^ j1)] = a[j1];
    }
    __syncthreads();
    // ---- pass 3: the 32-point transform of row (k1, j1): radix 2 (wave-uniform half c) then 16; bin k1 + 16 j1 + 256 c + 512 d
    {
      const int j1 = tid & 15, k1 = (tid >> 4) & 15, c = tid >> 8, jj = j1 ^ ((k1 & 1) << 4), rb = k1 * 512 + 32 * j1;
      if (c == 0) {
#pragma unroll
        for (int b = 0; b < 16; b++) a[b] = cadd(x[rb + (b ^ jj)], x[rb + ((b + 16) ^ jj)]);
      } else {
        // W_32^b = exp(-2 pi i b / 32)
        const float2 w32[16] = {{1.f, 0.f}, {0.98078528040323043f, -0.19509032201612825f}, {0.92387953251128674f, -0.38268343236508977f},
                                {0.83146961230254524f, -0.55557023301960218f}, {0.70710678118654752f, -0.70710678118654752f},
                                {0.55557023301960218f, -0.83146961230254524f}, {0.38268343236508977f, -0.92387953251128674f},
                                {0.19509032201612825f, -0.98078528040323043f}, {0.f, -1.f}, {-0.19509032201612825f, -0.98078528040323043f},
                                {-0.38268343236508977f, -0.92387953251128674f}, {-0.55557023301960218f, -0.83146961230254524f},
                                {-0.70710678118654752f, -0.70710678118654752f}, {-0.83146961230254524f, -0.55557023301960218f},
                                {-0.92387953251128674f, -0.38268343236508977f}, {-0.98078528040323043f, -0.19509032201612825f}};
#pragma unroll
        for (int b = 0; b < 16; b++) {
          const float2 dlt = csub(x[rb + (b ^ jj)], x[rb + ((b + 16) ^ jj)]);
          a[b] = b ? cmul(dlt, w32[b]) : dlt;
        }
      }
      dft16(a);
      __syncthreads();                                           // every row has been read
      const int base2 = (k1 ^ j1) + 16 * j1 + 256 * c;           // s8_swz2 of the shifted bin: out[b] = X[(b - N/2) mod N]
#pragma unroll
      for (int d = 0; d < 16; d++) x[base2 + 512 * (d ^ 8)] = a[d];
    }
    }
    if (more && tid < 128) s8_fill_ptab(ptab + (par ^ 1) * 128, mn, tid);
    __syncthreads();
    auto X = [&](int b) -> float2 { return x[s8_swz2(b)]; };
    if (TAPS && fft_tap) {
#pragma unroll
      for (int i = 0; i < 16; i++) { const int b = tid + i * S8_T; fft_tap[(size_t)s * N + b] = X(b); }
    }
    if (last) break;
    if (S8_EXP & 1) {
      if (more) {
        const long long low = (long long)(st->call0 + s_next) * (N + cp) + mn.cp_start - N + 1;
#pragma unroll
        for (int i = 0; i < 16; i++) vin[i] = iq[low + tid + i * S8_T];
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
      float sum = 0.f;
      for (int j = sub; j < S8_NCP - 1; j += 16) {
        const float2 u = X(i + s_cpil[j + 1]), v = X(i + s_cpil[j]);
        const float dx = u.x - v.x, dy = u.y - v.y;
        sum += s_known[j] * (dx * dx + dy * dy);
      }
      for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      if (sub == 0) s_cfo[cand] = sum;
    } else {
      const int q = tid - 256, j = q & 15, pat = (q >> 4) & 3, cg = q >> 6;
#pragma unroll
      for (int cc = 0; cc < 4; cc++) {
        const int cand = cg * 4 + cc;
        float cr = 0.f, ci = 0.f;
        if (j < 10) { const float2 v = X(zl - 8 + cand + 3 * pat + 12 * j); cr = pat_ref * v.x; ci = -pat_ref * v.y; }   // ref * conj(v)
        for (int o = 8; o > 0; o >>= 1) { cr += __shfl_xor(cr, o); ci += __shfl_xor(ci, o); }
        if (j == 0) s_pat[cand * 4 + pat] = cr * cr + ci * ci;
      }
    }
    __syncthreads();
    int fo, mod;
    {
      float mx = 0.f; int best = 8;
#pragma unroll
      for (int c = 0; c < 16; c++) { const float v = s_cfo[c]; if (v > mx) { mx = v; best = c; } }
      float mp = 0.f; int bm = 0;
#pragma unroll
      for (int c = 0; c < 4; c++) { const float v = s_pat[best * 4 + c]; if (v > mp) { mp = v; bm = c; } }
      fo = __builtin_amdgcn_readfirstlane(best - 8); mod = __builtin_amdgcn_readfirstlane(bm);
    }
    const int xb = zl + fo;
    if (tid == 0) { SymInfo si; si.freq_offset = fo; si.mod_index = mod; si.cfc = 0.f; si.pad = 0; info[s] = si; }
    // the next symbol's samples start travelling now
    if (more && !(S8_EXP & 4)) {
      const long long low = (long long)(st->call0 + s_next) * (N + cp) + mn.cp_start - N + 1;
#pragma unroll
      for (int i = 0; i < 16; i++) vin[i] = iq[low + tid + i * S8_T];
    }
    if (mod != cur_mod) {                                         // workgroup-uniform; in lock only at a workgroup's first symbol
      cur_mod = mod;
      const size_t tb = (size_t)mod * S8_PAY;
#pragma unroll
      for (int it = 0; it < (S8_IT + 3) / 4; it++) tdp[it] = 0;
#pragma unroll
      for (int it = 0; it < S8_IT; it++) {
        const int i = tid + it * S8_T, ic = i < S8_PAY ? i : S8_PAY - 1;
        tcl[it] = (unsigned)T.pay_c[tb + ic] | ((unsigned)T.pay_Li[tb + ic] << 16);
        tdp[it >> 2] |= (unsigned)T.pay_d[tb + ic] << (8 * (it & 3));
      }
      np = mod == 0 ? T.np[0] : mod == 1 ? T.np[1] : mod == 2 ? T.np[2] : T.np[3];
      const uint16_t *pk = T.pil_k + (size_t)mod * DEMOD_NP;      // carrier | sign of its reference << 15
      est01 = (unsigned)pk[tid < np ? tid : 0] | ((unsigned)pk[tid + S8_T < np ? tid + S8_T : 0] << 16);
      if (tid < S8_NTPS) { const int q = mod * S8_NTPS + tid; tps_ld = (unsigned)T.tps_Li[q] | ((unsigned)T.tps_d[q] << 16); }
    }
    // LS gains at the estimation carriers (set_channel_gain :486-490)
    {
      const float amp = (float)(4.0 / 3.0);
      const int e0 = (int)(est01 & 0xffffu), e1 = (int)(est01 >> 16);
      if (tid < np) gtab[tid] = cdiv(make_float2((e0 & 0x8000) ? -amp : amp, 0.f), X(xb + (e0 & 0x7fff)));
      if (tid + S8_T < np) gtab[tid + S8_T] = cdiv(make_float2((e1 & 0x8000) ? -amp : amp, 0.f), X(xb + (e1 & 0x7fff)));
    }
    __syncthreads();
    // interpolation (:617-642, the constant 11 of :625) + equalise (:1111-1114) + demap
    auto gain = [&](int Li, int dj) -> float2 {
      const float2 gl = gtab[Li], gr = gtab[Li + 1];
      const float k11 = 1.0f / 11.0f, tx = (gr.x - gl.x) * k11, ty = (gr.y - gl.y) * k11, j = (float)dj;
      return make_float2(gl.x + tx * j, gl.y + ty * j);
    };
    {
      uint8_t *lab = labels + (size_t)s * S8_PAY;
      bool slow = false;
#pragma unroll
      for (int it = 0; it < S8_IT; it++) {
        const int i = tid + it * S8_T;
        if (i < S8_PAY && !(S8_EXP & 8)) {
          const float2 e = cmul(X(xb + (int)(tcl[it] & 0xffffu)), gain((int)(tcl[it] >> 16), (int)((tdp[it >> 2] >> (8 * (it & 3))) & 0xffu)));
          if (TAPS && eq_tap) eq_tap[(size_t)s * S8_PAY + i] = e;
          const int f = demap_fast(e, pts, label_of, ip);
          slow |= f < 0;
          lab[i] = (uint8_t)f;
        }
      }
      // samples outside the range of the 4-candidate search (never on a locked signal): the exhaustive search, outside the unrolled loop
      if (__any(slow)) {
#pragma unroll 1
        for (int it = 0; it < S8_IT; it++) {
          const int i = tid + it * S8_T;
          if (i >= S8_PAY) break;
          const size_t tb = (size_t)mod * S8_PAY + i;
          const float2 e = cmul(X(xb + T.pay_c[tb]), gain(T.pay_Li[tb], T.pay_d[tb]));
          if (demap_fast(e, pts, label_of, ip) < 0) lab[i] = (uint8_t)demap_all(e, pts, ip.csize);
        }
      }
    }
    if (tid < S8_NTPS)    // equalised TPS carriers (process_tps_data :929-931)
      tpsval[(size_t)s * S8_NTPS + tid] = cmul(X(xb + tps_c), gain((int)(tps_ld & 0xffffu), (int)(tps_ld >> 16)));
    if (!more) break;
    s = s_next; m = mn; par ^= 1;
  }
}

}  // namespace dvbt
