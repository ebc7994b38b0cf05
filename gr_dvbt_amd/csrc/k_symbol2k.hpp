// k_symbol2k.hpp -- the per-OFDM-symbol kernel of the segment path for the 2k mode (A1 tail + A2 + A3 + A4): the design of k_symbol8k.hpp
// (persistent workgroups, symbols from an atomic counter, first FFT pass on the registers the samples arrive in, natural-order last pass,
// XOR-swizzled LDS, packed complex arithmetic, grid-cell demapper) with FOUR symbols per workgroup: a 2k symbol is 128 threads x 16 samples, so
// the 512 threads of a workgroup carry four consecutive symbols through the same barriers, each quarter with its own 16 KB image, gain table,
// phasor tables and search results.  FFT: 2048 = 16 x 8 x 16 (radix 16 on the registers, two radix-8 butterflies per thread, radix 16 per row).
// Reference lines as in k_symbol8k.hpp.
#pragma once
#include "k_symbol8k.hpp"

namespace dvbt {

constexpr int S2_N = 2048, S2_T = 128, S2_Q = 4, S2_PAY = 1512, S2_NCP = 45, S2_NTPS = 17, S2_ZL = 172, S2_NP = 192;
constexpr int S2_IT = (S2_PAY + S2_T - 1) / S2_T;             // payload carriers per thread (12)
constexpr int S2_PT = 104;                                    // phasor table entries per symbol
#ifndef S2_TOP_N
#define S2_TOP_N 7
#endif
constexpr int S2_TOP = S2_TOP_N;                              // samples per thread requested at the top of the iteration (see S8_TOP)
constexpr size_t S2_SLOT_BYTES = (size_t)S2_N * 8 + S2_NP * 8 + 2 * S2_PT * 8 + 16 * 4 + 64 * 4;       // per symbol of the four: image, gains, phasor tables x 2, search results
constexpr size_t S2_LDS_BYTES = S2_Q * S2_SLOT_BYTES + 64 * 8 + 64 + 48 * 4 + 48 * 2 + 16 + 40 * 4;
static_assert(S2_LDS_BYTES <= 81920, "two workgroups per CU");

// first two passes: a = k1 * 128 + (index inside the 128-point sub-transform k1), its low four bits XORed with k1 and bit 4 with k1's lowest
// bit: the strided reads of the radix-8 pass (16 x 2 sub-transforms per lane group) and of the last pass (16 sub-transforms x 2 rows) land on 32
// distinct bank pairs
__device__ __forceinline__ int s2_swz1(int k1, int idx) { return k1 * 128 + (idx ^ (k1 & 15) ^ ((k1 & 1) << 4)); }

// phasor tables of one symbol: [0,16) S_A(i) = expj(128 i incA), [16,32) S_B, [32,36) expj(thA + 32 a incA), [36,40) the same for B, [40,72)
// expj(b incA), [72,104) for B; sample n = t + 128 i of piece X has phase th_X + n inc_X
__device__ __forceinline__ void s2_fill_ptab(float2 *pt, const SymMeta &m, int t)
{
  const double thA = (double)m.ph_base + m.incA, thB = (double)m.ph_base + (double)m.sw * (m.incA - m.incB) + m.incB;
  double ang;
  if (t < 16) ang = 128.0 * t * m.incA;
  else if (t < 32) ang = 128.0 * (t - 16) * m.incB;
  else if (t < 36) ang = thA + 32.0 * (t - 32) * m.incA;
  else if (t < 40) ang = thB + 32.0 * (t - 36) * m.incB;
  else if (t < 72) ang = (double)(t - 40) * m.incA;
  else ang = (double)(t - 72) * m.incB;
  float sn, cs; sincosf(wrap_pi(ang), &sn, &cs);
  pt[t] = make_float2(cs, sn);
}

// 8-point forward DFT in registers: a[j], j = time index, becomes A[k]: radix 2 (with W8^j on the odd branch), then two radix-4 butterflies
__device__ __forceinline__ void s8_dft8(v2f (&a)[8], const S8Roots &R)
{
  const v2f u0 = a[0] + a[4], u1 = a[1] + a[5], u2 = a[2] + a[6], u3 = a[3] + a[7];
  const v2f v0 = a[0] - a[4], v1 = s8_mul_h_mh(a[1] - a[5], R.H), v2 = s8_mul_mi(a[2] - a[6]), v3 = s8_mul_mh_mh(a[3] - a[7], R.H);   // W8^1 = h(1-i), W8^2 = -i, W8^3 = -h(1+i)
  s8_bfly4(u0, u1, u2, u3, a[0], a[2], a[4], a[6]);
  s8_bfly4(v0, v1, v2, v3, a[1], a[3], a[5], a[7]);
}
__device__ __forceinline__ void s8_twiddle8(v2f (&a)[8], v2f w1)
{
  const v2f w2 = s8_cmul(w1, w1), w3 = s8_cmul(w2, w1), w4 = s8_cmul(w2, w2), w5 = s8_cmul(w4, w1), w6 = s8_cmul(w3, w3), w7 = s8_cmul(w4, w3);
  a[1] = s8_cmul(a[1], w1); a[2] = s8_cmul(a[2], w2); a[3] = s8_cmul(a[3], w3); a[4] = s8_cmul(a[4], w4);
  a[5] = s8_cmul(a[5], w5); a[6] = s8_cmul(a[6], w6); a[7] = s8_cmul(a[7], w7);
}
__device__ __forceinline__ float s8_oct_sum(float v) { v += s8_dpp<0xB1>(v); v += s8_dpp<0x4E>(v); v += s8_dpp<0x141>(v); return v; }   // 8 lanes: quad_perm x 2, row_half_mirror
__device__ __forceinline__ v2f s2_sample(s8_i4 rsrc, int i, int t) { return s8_raw_buffer_load_v2f32(rsrc, t * 8, i * S2_T * 8, 0); }

template <bool TAPS, bool DRIFT> __global__ __launch_bounds__(S2_T * S2_Q, 4) void symbol2k_kernel(const float2 *__restrict__ iq_, FrontParams p, const RxState *st,
                                                           const SymMeta *__restrict__ meta, const float2 *__restrict__ tw, float2 *__restrict__ acq_tap,
                                                           float2 *__restrict__ fft_tap, DemodTables T, float2 *__restrict__ eq_tap,
                                                           float2 *__restrict__ tpsval, SymInfo *__restrict__ info, InnerParams ip,
                                                           const float2 *__restrict__ points, const unsigned char *__restrict__ label_tab,
                                                           uint8_t *__restrict__ labels, int *__restrict__ ticket,
                                                           const float *__restrict__ drift, const int *__restrict__ drift_flags, float *__restrict__ csi_tap)
{
  if ((drift_flags[1] != 0) != DRIFT) return;                   // see symbol8k_kernel
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid0 = threadIdx.x, g = tid0 >> 7, t0 = tid0 & 127, cp = p.cp;          // g: which of the workgroup's four symbols
  unsigned char *slot = smem_raw + (size_t)g * S2_SLOT_BYTES;
  v2f *x = reinterpret_cast<v2f *>(slot);
  v2f *gtab = x + S2_N;                                          // LS gains at the estimation carriers
  float2 *ptab = reinterpret_cast<float2 *>(gtab + S2_NP);       // [2][S2_PT] phasor tables, this symbol's and the next one's
  float *s_cfo = reinterpret_cast<float *>(ptab + 2 * S2_PT);    // 16 candidate offsets
  float *s_pat = s_cfo + 16;                                     // [16 candidates][4 patterns]
  float2 *pts = reinterpret_cast<float2 *>(smem_raw + S2_Q * S2_SLOT_BYTES);
  unsigned char *label_of = reinterpret_cast<unsigned char *>(pts + 64);
  float *s_known = reinterpret_cast<float *>(label_of + 64);     // 48
  short *s_cpil = reinterpret_cast<short *>(s_known + 48);       // 48
  int *s_tkt = reinterpret_cast<int *>(s_cpil + 48);             // the group of four symbols this workgroup takes next
  float *s_pref = reinterpret_cast<float *>(s_tkt + 4);          // reference values of the first ten scattered pilots of the four patterns
  const v2f *iq = reinterpret_cast<const v2f *>(iq_);
  constexpr int N = S2_N, zl = S2_ZL;
  const int nsym = st->n_symbols, call0 = st->call0;
  if (tid0 == 0) s_tkt[0] = atomicAdd(ticket, 1);
  __syncthreads();
  int grp = __builtin_amdgcn_readfirstlane(s_tkt[0]);
  if (grp * S2_Q >= nsym) return;

  if (tid0 < 64) { pts[tid0] = points[tid0]; label_of[tid0] = label_tab[tid0]; }
  if (tid0 < S2_NCP) { s_cpil[tid0] = T.cpilot[tid0]; if (tid0 < S2_NCP - 1) s_known[tid0] = T.known_diff[tid0]; }
  const v2f w1A = s8_v(tw[t0]), w1B = s8_v(tw[16 * (t0 & 15)]);  // W_2048^n2 (first pass), W_128^m2 (second pass)
  const int tps_c = t0 < S2_NTPS ? T.tps[t0] : 0;
  if (tid0 < 40) s_pref[tid0] = T.pilot_ref[3 * (tid0 / 10) + 12 * (tid0 % 10)];
  const float half_n = 0.5f * (float)ip.nlev, top = (float)ip.nlev - 0.5f;
  int cur_mod = 0, s_prev = grp * S2_Q + g;

  // a quarter whose symbol lies beyond the segment works on a copy of the last symbol and stores nothing
  int s = grp * S2_Q + g;
  bool act = s < nsym;
  int sc = act ? s : nsym - 1;
  SymMeta m = meta[sc];
  v2f vin[16];
  {
    const long long low = (long long)(call0 + sc) * (N + cp) + m.cp_start - N + 1;
    const s8_i4 rs = s8_rsrc_lim(iq, low, p.avail);
#pragma unroll
    for (int i = S2_TOP; i < 16; i++) vin[i] = s2_sample(rs, i, t0);
  }
  if (t0 < S2_PT) s2_fill_ptab(ptab, m, t0);
  int par = 0;
  __syncthreads();

  for (;;) {
    int t = t0;
    asm volatile("" : "+v"(t));                                   // see k_symbol8k.hpp: nothing derived from the thread index is hoisted out of the loop
    const S8Roots R = s8_roots();
    int tkt = 0;
    if (tid0 == 0) tkt = atomicAdd(ticket, 1);
    const bool outp = act && !(!p.keep_last && s + 1 >= nsym);    // no output for the last item (the reference's demod consumes n+1 items)
    {
      const long long low = (long long)(call0 + sc) * (N + cp) + m.cp_start - N + 1;
      const s8_i4 rs = s8_rsrc_lim(iq, low, p.avail);
#pragma unroll
      for (int i = 0; i < S2_TOP; i++) vin[i] = s2_sample(rs, i, t);
    }
    // ---- A1 tail: derotate
    v2f a[16];
    {
      const v2f *pt = reinterpret_cast<const v2f *>(ptab) + par * S2_PT;
      const v2f PA = s8_cmul(pt[32 + (t >> 5)], pt[40 + (t & 31)]), PB = s8_cmul(pt[36 + (t >> 5)], pt[72 + (t & 31)]);
      const int sw = (m.sw >= 0 && m.sw < N + cp) ? m.sw : 0x7fffffff;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int i = (k + S2_TOP) & 15, n = t + i * S2_T;
        const bool pieceB = n + 1 > sw;
        const v2f P = pieceB ? PB : PA;
        a[i] = s8_cmul(s8_cmul(P, pt[(pieceB ? 16 : 0) + i]), vin[i]);
        if (DRIFT) {   // x (1 + i delta) of the sample's 32-sample block (k_drift.hpp); sample n = t + 128 i lies in block (t >> 5) + 4 i
          const float dl = drift[(size_t)sc * (S2_N / 32) + (t >> 5) + 4 * i];
          a[i] = (v2f){__builtin_fmaf(-dl, a[i].y, a[i].x), __builtin_fmaf(dl, a[i].x, a[i].y)};
        }
        if (TAPS && acq_tap && act) acq_tap[(size_t)s * N + n] = s8_f(a[i]);
      }
    }
    // ---- A2, pass 1: n = n2 + 128 n1 -> Y[k1][n2] W_2048^(n2 k1)
    s8_dft16(a, R);
    { v2f wA = w1A; asm volatile("" : "+v"(wA), "+v"(a[15]));
      s8_twiddle16(a, wA); }
    __syncthreads();                                             // the previous symbols' readers of x are done
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) x[s2_swz1(k1, t)] = a[k1];
    __syncthreads();
    // ---- pass 2: the 128-point transform of row k1, n2 = 16 m1 + m2 -> Z[k1][j1][m2] W_128^(m2 j1), radix 8, two rows per thread, in place
    {
      const int m2 = t & 15;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int k1 = (t >> 4) + 8 * h;
        v2f b[8];
#pragma unroll
        for (int m1 = 0; m1 < 8; m1++) b[m1] = x[s2_swz1(k1, 16 * m1 + m2)];
        s8_dft8(b, R);
        { v2f wB = w1B; asm volatile("" : "+v"(wB), "+v"(b[7]));
          s8_twiddle8(b, wB); }
#pragma unroll
        for (int j1 = 0; j1 < 8; j1++) x[s2_swz1(k1, 16 * j1 + m2)] = b[j1];
      }
    }
    __syncthreads();
    // ---- pass 3: the 16-point transform of row (k1, j1); bin k1 + 16 j1 + 128 j2
    {
      const int k1 = t & 15, j1 = t >> 4;
#pragma unroll
      for (int m2 = 0; m2 < 16; m2++) a[m2] = x[s2_swz1(k1, 16 * j1 + m2)];
      s8_dft16(a, R);
      __syncthreads();                                           // every row has been read
#pragma unroll
      for (int j2 = 0; j2 < 16; j2++) x[s8_swz2(k1 + 16 * j1 + 128 * (j2 ^ 8))] = a[j2];      // shifted: out[b] = X[(b - N/2) mod N]
    }
    if (tid0 == 0) s_tkt[1] = tkt;
    __syncthreads();
    const int grp_next = __builtin_amdgcn_readfirstlane(s_tkt[1]);
    const bool more = grp_next * S2_Q < nsym;
    const int s_next = grp_next * S2_Q + g;
    const bool act_next = s_next < nsym;
    const int sc_next = act_next ? s_next : nsym - 1;
    SymMeta mn = m;
    if (more) mn = meta[sc_next];
    auto X = [&](int b) -> v2f { return x[s8_swz2(b)]; };
    unsigned tcl[S2_IT];
    unsigned est0 = 0, est1 = 0, tps_l = 0, tps_dd = 0; int np = 0;
    auto load_rows = [&](int md) {
      const uint32_t *pp = T.pay_pack + (size_t)md * S2_PAY;
#pragma unroll
      for (int it = 0; it < S2_IT; it++) { const int i = t + it * S2_T; tcl[it] = pp[i < S2_PAY ? i : S2_PAY - 1]; }
      np = md == 0 ? T.np[0] : md == 1 ? T.np[1] : md == 2 ? T.np[2] : T.np[3];
      const uint16_t *pk = T.pil_k + (size_t)md * DEMOD_NP;
      est0 = pk[t < np ? t : 0]; est1 = pk[t + S2_T < np ? t + S2_T : 0];
      if (t < S2_NTPS) { const int q = md * S2_NTPS + t; tps_l = T.tps_Li[q]; tps_dd = T.tps_d[q]; }
    };
    const int pred = (cur_mod + (s - s_prev)) & 3;
    load_rows(pred);
    if (TAPS && fft_tap && act) {
#pragma unroll
      for (int i = 0; i < 16; i++) { const int b = t + i * S2_T; fft_tap[(size_t)s * N + b] = s8_f(X(b)); }
    }

    // ---- A3: integer CFO (16 candidate shifts x 44 pilot pairs, 8 lanes per candidate), then the symbol index mod 4 for every candidate shift
    // (first 10 scattered pilots of each pattern, two lanes per (candidate, pattern))
    {
      const int cand = t >> 3, sub = t & 7, i = zl - 8 + cand;
      constexpr int NJ = (S2_NCP - 1 + 7) / 8;                    // 44 pairs: 6 rounds of 8 lanes, the last one half empty
      float sum = 0.f;
      int c0[NJ], c1[NJ]; v2f u[NJ], v[NJ];
#pragma unroll
      for (int k = 0; k < NJ; k++) { const int j = sub + 8 * k < S2_NCP - 1 ? sub + 8 * k : 0; c0[k] = s_cpil[j]; c1[k] = s_cpil[j + 1]; }
#pragma unroll
      for (int k = 0; k < NJ; k++) { u[k] = X(i + c1[k]); v[k] = X(i + c0[k]); }
#pragma unroll
      for (int k = 0; k < NJ; k++) {
        const int j = sub + 8 * k;
        const v2f d = u[k] - v[k];
        sum += (j < S2_NCP - 1 ? s_known[j < S2_NCP - 1 ? j : 0] : 0.f) * (d.x * d.x + d.y * d.y);
      }
      sum = s8_oct_sum(sum);
      if (sub == 0) s_cfo[cand] = sum;
    }
    {
      const int combo = t >> 1, cand = combo >> 2, pat = combo & 3, half = t & 1;
      float cr = 0.f, ci = 0.f;
#pragma unroll
      for (int j = 0; j < 5; j++) { const v2f v = X(zl - 8 + cand + 3 * pat + 12 * (5 * half + j)); const float r = s_pref[10 * pat + 5 * half + j]; cr += r * v.x; ci -= r * v.y; }   // ref * conj(v)
      cr += s8_dpp<0xB1>(cr); ci += s8_dpp<0xB1>(ci);
      if (half == 0) s_pat[cand * 4 + pat] = cr * cr + ci * ci;
    }
    if (more && t < S2_PT) s2_fill_ptab(ptab + (par ^ 1) * S2_PT, mn, t);
    __syncthreads();
    int fo, mod;
    {
      const int l = t & 63;
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
    if (t == 0 && outp) { SymInfo si; si.freq_offset = fo; si.mod_index = mod; si.cfc = 0.f; si.pad = 0; info[s] = si; }
    if (more) {
      const long long low = (long long)(call0 + sc_next) * (N + cp) + mn.cp_start - N + 1;
      const s8_i4 rs = s8_rsrc_lim(iq, low, p.avail);
#pragma unroll
      for (int i = S2_TOP; i < 16; i++) vin[i] = s2_sample(rs, i, t);
    }
    if (mod != pred) load_rows(mod);
    cur_mod = mod; s_prev = s;
    {
      const float amp = (float)(4.0 / 3.0);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int e = (int)(h ? est1 : est0), r = t + h * S2_T;
        if (r < np) {
          const v2f v = X(xb + (e & 0x7fff));
          const float q = ((e & 0x8000) ? -amp : amp) * __builtin_amdgcn_rcpf(v.x * v.x + v.y * v.y);
          gtab[r] = (v2f){q * v.x, -q * v.y};
        }
      }
    }
    __syncthreads();
    auto equalise = [&](int c, int Li, int dj) -> v2f {
      const v2f gl = gtab[Li], gr = gtab[Li + 1];
      const float k11 = 1.0f / 11.0f, j = (float)dj;
      const v2f tt = (gr - gl) * k11;
      return s8_cmul(X(xb + c), (v2f){__builtin_fmaf(tt.x, j, gl.x), __builtin_fmaf(tt.y, j, gl.y)});
    };
    {
      uint8_t *lab = labels + (size_t)s * S2_PAY;
      bool slow = ip.hshift != 0.f;                              // hierarchical constellations: every carrier through the candidate search on the shifted grid
#pragma unroll
      for (int it = 0; it < S2_IT; it++) {
        const int i = t + it * S2_T;
        if ((it < S2_PAY / S2_T || i < S2_PAY) && outp) {
          const v2f e = equalise((int)(tcl[it] & 0x1fffu), (int)((tcl[it] >> 13) & 0x3ffu), (int)(tcl[it] >> 23));
          if (TAPS && eq_tap) eq_tap[(size_t)s * S2_PAY + i] = s8_f(e);
          if (TAPS && csi_tap) {   // channel state of the carrier for the soft-decision path (k_soft.hpp)
            const int Li = (int)((tcl[it] >> 13) & 0x3ffu); const float j = (float)(tcl[it] >> 23);
            const v2f gl = gtab[Li], gr = gtab[Li + 1], t2 = (gr - gl) * (1.0f / 11.0f);
            const float gx = __builtin_fmaf(t2.x, j, gl.x), gy = __builtin_fmaf(t2.y, j, gl.y);
            csi_tap[(size_t)s * S2_PAY + i] = 1.0f / (gx * gx + gy * gy);
          }
          int idx;
          slow |= !s8_demap_cell(e, ip.inv_step, half_n, top, idx);
          lab[i] = label_of[idx & 63];
        }
      }
      if (__any(slow)) {
#pragma unroll 1
        for (int it = 0; it < S2_IT; it++) {
          const int i = t + it * S2_T;
          if (i >= S2_PAY || !outp) break;
          const unsigned w = T.pay_pack[(size_t)mod * S2_PAY + i];
          const v2f e = equalise((int)(w & 0x1fffu), (int)((w >> 13) & 0x3ffu), (int)(w >> 23));
          int f = demap_fast(s8_f(e), pts, label_of, ip);
          if (f < 0) f = demap_all(s8_f(e), pts, ip.csize);
          lab[i] = (uint8_t)f;
        }
      }
    }
    if (t < S2_NTPS && outp)
      tpsval[(size_t)s * S2_NTPS + t] = s8_f(equalise(tps_c, (int)tps_l, (int)tps_dd));
    if (!more) break;
    s = s_next; sc = sc_next; act = act_next; m = mn; par ^= 1;
  }
}

}  // namespace dvbt
