// k_symbol.hpp -- the per-OFDM-symbol kernel of the segment path: A1 tail (derotate, strip CP) + A2 (FFT, shift) +
// A3 (pilot engine, equaliser) + A4 (demapper) in one workgroup, the symbol resident in LDS from the first load of its
// baseband samples to the store of its 6048 one-byte labels.
#pragma once
#include "k_frontend.hpp"
#include "k_backend.hpp"

namespace dvbt {

// One workgroup per OFDM symbol: derotate + strip CP + FFT as derot_fft_kernel, then the pilot engine of
// demod_kernel on the spectrum while it is still in LDS -- the 64 KB item never travels to HBM and back.
// Differences to demod_kernel, both inside the float tolerance of the equalised-carrier tap:
//  * the common phasor of frequency_correction (:793-819) is not applied: it multiplies the pilots and the payload
//    alike, so it cancels in  x[c] * ref / x[pilot]  (and with it the only use of the NEXT symbol disappears);
//  * the LS gain of an estimation carrier is computed once (rank table) instead of once per carrier that uses it,
//    and the interpolation step (g[R]-g[L])/11 (:625) is a multiplication by 1/11.
// A4 rides along: every equalised carrier is demapped at once (demap_one: the reference's first strict minimum), so a
// symbol leaves the kernel as `payload` label bytes in carrier order; the equalised carriers themselves are written
// only when the EQ tap is enabled.
#ifndef SYM_EXP
#define SYM_EXP 0     // experiment builds only (tools/sym_attribution.sh; wrong output): 1 stop after the load + derotation, 2 no FFT, 4 no reorder,
                      // 8 no integer-CFO / pattern search, 16 no equaliser + demapper
#endif
constexpr int SYM_NCP_MAX = 192;           // continual pilots of a mode (177 in 8k)
inline size_t fused_lds_bytes_host(int N) { return (size_t)(N + N / 32 + N / 128 + 128 + DEMOD_NP) * 8 + 128 + 64 * 8 + 64 + SYM_NCP_MAX * 6; }

__global__ __launch_bounds__(FFT_THREADS, 4) void derot_fft_demod_kernel(const float2 *__restrict__ iq, FrontParams p, const RxState *st,
                                                             const SymMeta *__restrict__ meta, const float2 *__restrict__ tw,
                                                             const uint16_t *__restrict__ perm, float2 *__restrict__ acq_tap,
                                                             float2 *__restrict__ fft_tap, DemodTables T, float2 *__restrict__ eq_tap,
                                                             float2 *__restrict__ tpsval, SymInfo *__restrict__ info, InnerParams ip,
                                                             const float2 *__restrict__ points, const unsigned char *__restrict__ label_tab,
                                                             uint8_t *__restrict__ labels)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2 *x = reinterpret_cast<float2 *>(smem_raw);
  const int s = blockIdx.x;
  const int nsym = st->n_symbols;
  if (s >= nsym) return;
  const bool last = !p.keep_last && s + 1 >= nsym;                               // no output for the last item (the reference's demod consumes n+1 items)
  if (last && !fft_tap && !acq_tap) return;
  const int N = p.N, cp = p.cp, tid = threadIdx.x, zl = p.zl;
  float2 *tw_c = x + (N + N / 32), *tw_f = tw_c + N / 128, *gtab = tw_f + 128;
  float *s_sum = reinterpret_cast<float *>(gtab + DEMOD_NP);
  int *s_i = reinterpret_cast<int *>(s_sum + 16);
  float2 *pts = reinterpret_cast<float2 *>(s_i + 16);
  unsigned char *label_of = reinterpret_cast<unsigned char *>(pts + 64);
  float *s_known = reinterpret_cast<float *>(label_of + 64);
  short *s_cpil = reinterpret_cast<short *>(s_known + SYM_NCP_MAX);
  if (tid < 64) { pts[tid] = points[tid]; label_of[tid] = label_tab[tid]; }
  if (tid < p.n_cp) { s_cpil[tid] = T.cpilot[tid]; if (tid < p.n_cp - 1) s_known[tid] = T.known_diff[tid]; }
  for (int i = tid; i < N / 128; i += FFT_THREADS) tw_c[i] = tw[i * 128];
  if (tid < 128) tw_f[tid] = tw[tid];
  const SymMeta m = meta[s];
  const long long low = (long long)(st->call0 + s) * (N + cp) + m.cp_start - N + 1;
  const bool rot = (m.incA != 0.0) || (m.incB != 0.0) || (m.ph_base != 0.f);
  // derot[n] = expj(phase after n+1 increments) (:285-309,:527-534).  The phase is piecewise linear in n (increment
  // incA up to the switch position sw, incB after it), so expj(phase(tid + T*i)) = P(tid) * S(i) with one sincos per
  // thread and piece (P) and a wave-uniform table of the T-sample steps (S) instead of one sincos per sample.
  const bool has_sw = m.sw >= 0 && m.sw < N + cp;
  float2 PA = make_float2(1.f, 0.f), PB = PA;
  float2 *stab = gtab;                                            // [2][N / FFT_THREADS], free until the pilot engine runs
  const int nstep = N / FFT_THREADS;
  float2 vin[8192 / FFT_THREADS];
#pragma unroll
  for (int i = 0; i < 8192 / FFT_THREADS; i++) vin[i] = iq[low + ((tid + i * FFT_THREADS) & (N - 1))];     // unconditional (wraps for N < 8192): loads must not sit behind branches
  if (rot) {
    const double thA = (double)m.ph_base + m.incA, thB = (double)m.ph_base + (double)m.sw * (m.incA - m.incB) + m.incB;
    if (tid < 2 * nstep) {
      const int i = tid % nstep;
      const float ph = wrap_pi((double)FFT_THREADS * i * (tid < nstep ? m.incA : m.incB));
      float sn, cs; sincosf(ph, &sn, &cs); stab[tid] = make_float2(cs, sn);
    }
    float sn, cs;
    sincosf(wrap_pi(thA + tid * m.incA), &sn, &cs); PA = make_float2(cs, sn);
    sincosf(wrap_pi(thB + tid * m.incB), &sn, &cs); PB = make_float2(cs, sn);
    __syncthreads();
  }
  constexpr int NSTEP_MAX = 8192 / FFT_THREADS;                   // every load of the symbol is in flight before the first use
#pragma unroll
  for (int i = 0; i < NSTEP_MAX; i++) if (i < nstep) {
    const int n = tid + i * FFT_THREADS;
    float2 v = vin[i];
    if (rot) {
      const bool pieceB = has_sw && n + 1 > m.sw;
      v = cmul(cmul(pieceB ? PB : PA, stab[(pieceB ? nstep : 0) + i]), v);
    }
    x[fpad(n)] = v;
    if (acq_tap) acq_tap[(size_t)s * N + n] = v;
  }
  __syncthreads();
  if (SYM_EXP & 1) { if (x[fpad(tid)].x == 123.f) labels[s] = 1; return; }
  if (!(SYM_EXP & 2)) fft_dif_lds(x, N, tw_c, tw_f, tid);
  if (!(SYM_EXP & 4)) {   // digit-reversed -> natural, fft-shifted order, in place through registers: x[b] = X[(b - N/2) mod N]
    constexpr int NR = 8192 / FFT_THREADS;
    unsigned short pos[NR]; float2 r[NR];
#pragma unroll
    for (int i = 0; i < NR; i++) pos[i] = perm[(tid + i * FFT_THREADS) & (N - 1)];      // dvbt_tables.hpp::fft_out_perm; all loads in flight at once
#pragma unroll
    for (int i = 0; i < NR; i++) {
      r[i] = x[fpad(pos[i])];       // (folding the last radix-2 stage into this gather was measured 35 % slower: two scattered reads per bin)
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NR; i++) { const int b = tid + i * FFT_THREADS; if (i < nstep) { x[fpad(b)] = r[i]; if (fft_tap) fft_tap[(size_t)s * N + b] = r[i]; } }
    __syncthreads();
  }
  if (last) return;
  auto X = [&](int b) -> float2 { return x[fpad(b)]; };

  // integer CFO: process_cpilot_data :715-744 -- 16 candidate shifts x (n_cp-1) pilot pairs
  if (!(SYM_EXP & 8)) {
    const int cand = (tid >> 4) & 15, sub = tid & 15, i = zl - 8 + cand;
    float sum = 0.f;
    for (int j = sub; j < p.n_cp - 1 && tid < 256; j += 16) {
      const float2 a = X(i + s_cpil[j + 1]), b = X(i + s_cpil[j]);
      const float dx = a.x - b.x, dy = a.y - b.y;
      sum += s_known[j] * (dx * dx + dy * dy);
    }
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (sub == 0 && tid < 256) s_sum[cand] = sum;
  }
  __syncthreads();
  if (tid == 0) {
    float mx = 0.f; int start = zl;
    for (int c = 0; c < 16 && !(SYM_EXP & 8); c++) if (s_sum[c] > mx) { mx = s_sum[c]; start = zl - 8 + c; }
    s_i[0] = start - zl;
  }
  __syncthreads();
  const int fo = s_i[0], xb = zl + fo;
  // symbol index mod 4: process_spilot_data :549-582 -- first 10 scattered pilots of each pattern
  if (tid < 64) {
    const int pat = tid >> 4, j = tid & 15;
    float cr = 0.f, ci = 0.f;
    if (j < 10) {
      const int k = 3 * pat + 12 * j;
      const float2 v = X(xb + k);
      const float r = T.pilot_ref[k];
      cr = r * v.x; ci = -r * v.y;                // ref * conj(v)
    }
    for (int o = 8; o > 0; o >>= 1) { cr += __shfl_xor(cr, o); ci += __shfl_xor(ci, o); }
    if (j == 0) s_sum[pat] = cr * cr + ci * ci;
  }
  __syncthreads();
  if (tid == 0) {
    float mx = 0.f; int mod = 0;
    for (int c = 0; c < 4; c++) if (s_sum[c] > mx) { mx = s_sum[c]; mod = c; }
    s_i[1] = mod;
    SymInfo si; si.freq_offset = fo; si.mod_index = mod; si.cfc = 0.f; si.pad = 0;
    info[s] = si;
  }
  __syncthreads();
  const int mod = s_i[1];
  // the equaliser's table rows of this thread (carrier, bracketing ranks, distance) start travelling now
  constexpr int PAY_IT = (6048 + FFT_THREADS - 1) / FFT_THREADS;
  const int nit = (p.payload + FFT_THREADS - 1) / FFT_THREADS;
  unsigned short tc[PAY_IT], tl[PAY_IT], tr[PAY_IT]; unsigned char td[PAY_IT];
  {
    const size_t tb = (size_t)mod * p.payload;
#pragma unroll
    for (int it = 0; it < PAY_IT; it++) {
      const int i = tid + it * FFT_THREADS;
      const int ic = i < p.payload ? i : p.payload - 1;            // clamped, unconditional
      tc[it] = T.pay_c[tb + ic]; tl[it] = T.pay_Li[tb + ic]; tr[it] = T.pay_Ri[tb + ic]; td[it] = T.pay_d[tb + ic];
    }
  }
  // LS gains at the estimation carriers (set_channel_gain :486-490); pil_k: carrier | sign of its reference << 15
  {
    const uint16_t *pk = T.pil_k + (size_t)mod * DEMOD_NP;
    const int np = T.np[mod];
    const float amp = (float)(4.0 / 3.0);
    const int e0 = pk[tid < np ? tid : 0], e1 = pk[tid + FFT_THREADS < np ? tid + FFT_THREADS : 0];               // np <= 2 * FFT_THREADS
    if (tid < np) gtab[tid] = cdiv(make_float2((e0 & 0x8000) ? -amp : amp, 0.f), X(xb + (e0 & 0x7fff)));
    if (tid + FFT_THREADS < np) gtab[tid + FFT_THREADS] = cdiv(make_float2((e1 & 0x8000) ? -amp : amp, 0.f), X(xb + (e1 & 0x7fff)));
  }
  __syncthreads();
  // interpolation (:617-642, the constant 11 of :625) + equalise (:1111-1114)
  auto gain = [&](int Li, int Ri, int dj) -> float2 {
    const float2 gl = gtab[Li], gr = gtab[Ri];
    const float k11 = 1.0f / 11.0f, tx = (gr.x - gl.x) * k11, ty = (gr.y - gl.y) * k11, j = (float)dj;
    return make_float2(gl.x + tx * j, gl.y + ty * j);
  };
  {
    uint8_t *lab = labels + (size_t)s * p.payload;
    bool slow = false;
#pragma unroll
    for (int it = 0; it < PAY_IT; it++) {
      const int i = tid + it * FFT_THREADS;
      if (it < nit && i < p.payload && !(SYM_EXP & 16)) {
        const float2 e = cmul(X(xb + tc[it]), gain(tl[it], tr[it], td[it]));
        if (eq_tap) eq_tap[(size_t)s * p.payload + i] = e;
        const int f = demap_fast(e, pts, label_of, ip);
        slow |= f < 0;
        lab[i] = (uint8_t)f;
      }
    }
    // samples outside the range of the 4-candidate search (never on a locked signal): the exhaustive search, kept out of
    // the unrolled loop above; the carrier is simply equalised again
    if (__any(slow)) {
      const size_t tb = (size_t)mod * p.payload;
#pragma unroll 1
      for (int i = tid; i < p.payload; i += FFT_THREADS) {
        const float2 e = cmul(X(xb + T.pay_c[tb + i]), gain(T.pay_Li[tb + i], T.pay_Ri[tb + i], T.pay_d[tb + i]));
        if (demap_fast(e, pts, label_of, ip) < 0) lab[i] = (uint8_t)demap_all(e, pts, ip.csize);
      }
    }
  }
  if (tid < p.n_tps) {   // equalised TPS carriers (process_tps_data :929-931)
    const int c = T.tps[tid], q = mod * p.n_tps + tid;
    tpsval[(size_t)s * p.n_tps + tid] = cmul(X(xb + c), gain(T.tps_Li[q], T.tps_Ri[q], T.tps_d[q]));
  }
}

}  // namespace dvbt
