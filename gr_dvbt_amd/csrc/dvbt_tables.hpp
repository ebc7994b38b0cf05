// dvbt_tables.hpp -- host-side derived constants and lookup tables of the DVB-T RX path.
//
// Product code (independent of oracle/).  Each builder cites the reference lines whose
// behaviour it has to reproduce (paths relative to the gr-dvbt tree).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace dvbt {

// ETSI EN 300 744 table 7 / table 8, 2k mode (reference: lib/reference_signals_impl.cc:54-70).
// The 8k tables (:76-117) are these repeated with period 1704, plus the last continual pilot.
static const int kCpilot2k[45] = {0, 48, 54, 87, 141, 156, 192, 201, 255, 279, 282, 333, 432, 450, 483,
  525, 531, 618, 636, 714, 759, 765, 780, 804, 873, 888, 918, 939, 942, 969, 984, 1050, 1101, 1107,
  1110, 1137, 1140, 1146, 1206, 1269, 1323, 1377, 1491, 1683, 1704};
static const int kTps2k[17] = {34, 50, 209, 346, 413, 569, 595, 688, 790, 901, 1073, 1219, 1262, 1286,
  1469, 1594, 1687};

struct Dims {
  int constellation = 1, hierarchy = 0, code_rate = 0, guard = 0, mode = 0;
  int N = 0, cp = 0, Kmax = 0, K = 0, payload = 0, zl = 0;
  int m = 0, csize = 0, alpha = 1, k = 1, n = 2;
  float norm = 0.f;
  int n_cp = 0, n_tps = 0, n_sp = 0;
  int ntb = 5;                 // traceback depth in bytes, viterbi_decoder_impl.cc:95-124
  int plen = 2;                // puncture vector length 2k, viterbi_decoder_impl.cc:61-65
  uint8_t punct[16] = {0};
  uint8_t prefix[16] = {0};    // received bits before phase p within one period
  int fi_start = 3;            // demod_reference_signals_impl.cc:73-77
  int info_bits_per_symbol = 0;
  bool valid = false;
};

inline Dims make_dims(int constellation, int hierarchy, int code_rate, int guard, int mode)
{
  Dims d;
  if (constellation < 0 || constellation > 2 || code_rate < 0 || code_rate > 4 || guard < 0 || guard > 3 ||
      mode < 0 || mode > 1 || hierarchy < 0 || hierarchy > 3) return d;
  d.constellation = constellation; d.hierarchy = hierarchy; d.code_rate = code_rate; d.guard = guard; d.mode = mode;
  // dvbt_config.cc:105-125
  if (mode == 1) { d.Kmax = 6816; d.N = 8192; d.payload = 6048; d.n_cp = 177; d.n_tps = 68; d.n_sp = 568; }
  else           { d.Kmax = 1704; d.N = 2048; d.payload = 1512; d.n_cp = 45;  d.n_tps = 17; d.n_sp = 142; }
  d.K = d.Kmax + 1;
  d.zl = (int)std::ceil((d.N - d.K) / 2.0);
  d.m = constellation == 0 ? 2 : constellation == 2 ? 6 : 4;          // :126-148
  d.csize = 1 << d.m;
  static const int kk[5] = {1, 2, 3, 5, 7};                           // :172-192 (LP switch wins; HP==LP in every caller)
  d.k = kk[code_rate]; d.n = d.k + 1;
  static const int gd[4] = {32, 16, 8, 4};                            // :194-211
  d.cp = d.N / gd[guard];
  d.alpha = hierarchy == 2 ? 2 : hierarchy == 3 ? 4 : 1;              // :213-225
  double nrm;                                                         // :229-249
  if (d.m == 2) nrm = 1.0 / std::sqrt(2.0);
  else if (d.m == 6) nrm = d.alpha == 1 ? 1.0 / std::sqrt(42.0) : d.alpha == 2 ? 1.0 / std::sqrt(60.0) : 1.0 / std::sqrt(108.0);
  else nrm = d.alpha == 1 ? 1.0 / std::sqrt(10.0) : d.alpha == 2 ? 1.0 / std::sqrt(20.0) : 1.0 / std::sqrt(52.0);
  d.norm = (float)nrm;
  static const int tb[5] = {5, 9, 10, 15, 24};
  d.ntb = tb[code_rate];
  static const uint8_t P[5][14] = {{1, 1}, {1, 1, 0, 1}, {1, 1, 0, 1, 1, 0}, {1, 1, 0, 1, 1, 0, 0, 1, 1, 0},
                                   {1, 1, 0, 1, 0, 1, 0, 1, 1, 0, 0, 1, 1, 0}};
  d.plen = 2 * d.k;
  int acc = 0;
  for (int i = 0; i < d.plen; i++) { d.punct[i] = P[code_rate][i]; d.prefix[i] = (uint8_t)acc; acc += d.punct[i]; }
  d.fi_start = (constellation == 2 && mode == 1) ? 2 : 3;
  d.info_bits_per_symbol = d.payload * d.m * d.k / d.n;
  d.valid = true;
  return d;
}

inline std::vector<int> cpilot_table(const Dims &d)
{
  std::vector<int> v;
  if (d.mode == 0) { v.assign(kCpilot2k, kCpilot2k + 45); return v; }
  for (int rep = 0; rep < 4; rep++) for (int i = 0; i < 44; i++) v.push_back(kCpilot2k[i] + 1704 * rep);
  v.push_back(6816);
  return v;
}
inline std::vector<int> tps_table(const Dims &d)
{
  std::vector<int> v;
  for (int rep = 0; rep < (d.mode == 1 ? 4 : 1); rep++) for (int i = 0; i < 17; i++) v.push_back(kTps2k[i] + 1704 * rep);
  return v;
}

// w_k PRBS x^11+x^2+1, all-ones start (reference_signals_impl.cc:334-345); pilot reference value +-4/3
inline std::vector<float> pilot_ref_table(const Dims &d)
{
  std::vector<float> v(d.K);
  unsigned reg = (1u << 11) - 1;
  for (int k = 0; k < d.K; k++) {
    int w = reg & 1;
    v[k] = (float)(4 * 2 * (0.5 - w) / 3);
    unsigned nb = ((reg >> 2) ^ reg) & 1;
    reg = (reg >> 1) | (nb << 10);
  }
  return v;
}

// symbol interleaver permutation H(q), symbol_inner_interleaver_impl.cc:32-96
inline std::vector<uint16_t> symbol_H(const Dims &d)
{
  static const int p2k[] = {4, 3, 9, 6, 2, 8, 1, 5, 7, 0};
  static const int p8k[] = {7, 1, 4, 2, 9, 6, 8, 10, 0, 3, 11, 5};
  const int Nr = d.mode == 1 ? 13 : 11;
  const int *perm = d.mode == 1 ? p8k : p2k;
  std::vector<uint16_t> h;
  int reg = 0;
  for (int i = 0; i < d.N; i++) {
    if (i < 2) reg = 0;
    else if (i == 2) reg = 1;
    else {
      int nb = d.mode == 1 ? (reg ^ (reg >> 1) ^ (reg >> 4) ^ (reg >> 6)) & 1 : (reg ^ (reg >> 3)) & 1;
      reg = ((reg >> 1) | (nb << (Nr - 2))) & ((1 << Nr) - 1);
    }
    int r = 0;
    for (int b = 0; b < Nr - 1; b++) r |= ((reg >> b) & 1) << perm[b];
    int v = ((i & 1) << (Nr - 1)) + r;
    if (v < d.payload) h.push_back((uint16_t)v);
  }
  return h;
}

// Constellation table indexed by the bit label y0..y(m-1) (y0 = MSB of the label): what dvbt_demap_impl.cc:117-165 builds, from
// the closed form of ETSI EN 300 744 4.3.5 (checked there against the compiled reference): even bits belong
// to I, odd bits to Q; the first bit of an axis is its sign (1 = negative), the others are the Gray code of the level counted
// from the outside: |level| = alpha + 2 (L - 1 - g), L = levels per half axis, g = Gray-decoded index.  Returns re,im pairs.
inline std::vector<float> constellation_points(const Dims &d, float gain_in)
{
  const float scale = gain_in * d.norm;
  const int per_axis = d.m / 2, half_levels = 1 << (per_axis - 1);
  std::vector<float> pts(2 * d.csize);
  for (int label = 0; label < d.csize; label++)
    for (int axis = 0; axis < 2; axis++) {                        // 0: I (y0, y2, y4), 1: Q (y1, y3, y5)
      auto y = [&](int j) { return (label >> (d.m - 1 - (2 * j + axis))) & 1; };   // j-th bit of this axis
      int g = 0, acc = 0;
      for (int j = 1; j < per_axis; j++) { acc ^= y(j); g = (g << 1) | acc; }      // Gray -> binary, MSB first
      const int level = d.alpha + 2 * (half_levels - 1 - g);
      pts[2 * label + axis] = scale * (float)(y(0) ? -level : level);
    }
  return pts;
}

// carriers of scattered-pilot pattern s in ascending order (get_current_spilot/advance_spilot,
// reference_signals_impl.cc:466-504)
inline std::vector<int> spilot_list(const Dims &d, int s)
{
  std::vector<int> v;
  int size = d.n_sp + (s == 0 ? 1 : 0);
  for (int p = 0; p < size; p++) { int k = 3 * s + 12 * p; if (k > d.Kmax) break; v.push_back(k); }
  return v;
}

// For pattern s: the ascending payload carrier list (process_payload_data :1065-1106) and, per
// carrier of interest, the two estimation carriers (scattered U continual) that bracket it
// (process_spilot_data :597-642): gain(c) = g[L] + (c-L)*(g[R]-g[L])/11, L==c for a pilot.
struct PatternTables {
  std::vector<uint16_t> pay_c, pay_L, pay_R, tps_L, tps_R;
  // the same brackets as ranks into the sorted list of estimation carriers (pil_k), plus the distance c - L
  std::vector<uint16_t> pil_k, pay_Li, pay_Ri, tps_Li, tps_Ri;
  std::vector<uint8_t> pay_d, tps_d;
};
inline PatternTables pattern_tables(const Dims &d, int s)
{
  std::vector<int> cpl = cpilot_table(d), tps = tps_table(d), sp = spilot_list(d, s);
  std::vector<char> is_est(d.K, 0), is_tps(d.K, 0);
  for (int k : cpl) is_est[k] = 1;
  for (int k : sp) is_est[k] = 1;
  for (int k : tps) is_tps[k] = 1;
  std::vector<int> L(d.K), R(d.K);
  int last = 0;
  for (int k = 0; k < d.K; k++) { if (is_est[k]) last = k; L[k] = last; }
  last = d.Kmax;
  for (int k = d.Kmax; k >= 0; k--) { if (is_est[k]) last = k; R[k] = last; }
  PatternTables t;
  for (int k = 0; k < d.K; k++)
    if (!is_est[k] && !is_tps[k]) { t.pay_c.push_back((uint16_t)k); t.pay_L.push_back((uint16_t)L[k]); t.pay_R.push_back((uint16_t)R[k]); }
  for (int k : tps) { t.tps_L.push_back((uint16_t)L[k]); t.tps_R.push_back((uint16_t)R[k]); }
  std::vector<int> rank(d.K, 0);
  for (int k = 0; k < d.K; k++) if (is_est[k]) { rank[k] = (int)t.pil_k.size(); t.pil_k.push_back((uint16_t)k); }
  for (size_t i = 0; i < t.pay_c.size(); i++) {
    t.pay_Li.push_back((uint16_t)rank[t.pay_L[i]]); t.pay_Ri.push_back((uint16_t)rank[t.pay_R[i]]); t.pay_d.push_back((uint8_t)(t.pay_c[i] - t.pay_L[i]));
  }
  for (size_t i = 0; i < tps.size(); i++) {
    t.tps_Li.push_back((uint16_t)rank[t.tps_L[i]]); t.tps_Ri.push_back((uint16_t)rank[t.tps_R[i]]); t.tps_d.push_back((uint8_t)(tps[i] - t.tps_L[i]));
  }
  return t;
}

// FFT plan: DIF passes of radix 16 (then 4 and/or 2), in place, natural-order
// input.  After the last stage position p holds X[k(p)] with k the digit reversal of p.
// perm[b] = position holding the bin that the shifted output index b must receive:
// out[b] = X[(b - N/2) mod N]   (gr::fft::fft_vcc forward, shift=True; SURVEY C-2).
inline std::vector<int> fft_radices(int N)
{ // the pass sequence of k_frontend.hpp::fft_dif_lds: radix 16 while the span allows, then 4 and/or 2
  std::vector<int> r;
  int L = N;
  while (L >= 16) { r.push_back(16); L /= 16; }
  if (L >= 4) { r.push_back(4); L /= 4; }
  if (L == 2) r.push_back(2);
  return r;
}
inline std::vector<uint16_t> fft_out_perm(int N)
{
  std::vector<int> rad = fft_radices(N);
  std::vector<int> pos_of_k(N);
  for (int p = 0; p < N; p++) {
    int span = N, weight = 1, k = 0;
    for (int r : rad) { span /= r; int dgt = (p / span) % r; k += dgt * weight; weight *= r; }
    pos_of_k[k] = p;
  }
  std::vector<uint16_t> perm(N);
  for (int b = 0; b < N; b++) perm[b] = (uint16_t)pos_of_k[(b + N / 2) % N];
  return perm;
}
inline std::vector<float> fft_twiddles(int N)
{
  std::vector<float> tw(2 * N);
  for (int t = 0; t < N; t++) { double a = -2.0 * M_PI * t / N; tw[2 * t] = (float)std::cos(a); tw[2 * t + 1] = (float)std::sin(a); }
  return tw;
}

// energy dispersal xor pattern for one group of 8 packets (energy_descramble_impl.cc:46-67,143-163)
inline std::vector<uint8_t> energy_prbs()
{
  std::vector<uint8_t> seq(1504, 0);
  unsigned reg = 0xa9;
  auto clk8 = [&reg]() { int res = 0; for (int i = 0; i < 8; i++) { int fb = ((reg >> 13) ^ (reg >> 14)) & 1; reg = ((reg << 1) | fb) & 0x7fff; res = (res << 1) | fb; } return res; };
  int c = 0;
  for (int p = 0; p < 8; p++) { seq[c++] = 0; for (int k = 1; k < 188; k++) seq[c++] = (uint8_t)clk8(); clk8(); }
  return seq;
}

// ---- front of the flowgraph: stock gr::filter rational_resampler_ccc(64, 70, taps=None, fbw=None).
// Third-party (gr-filter / gr-fft of the GNU Radio 3.7 series, absent from the reference tree, version unpinned): the
// published algorithm is restated -- rational_resampler.py (gcd reduction, fractional_bw 0.4, design_filter),
// firdes.cc (compute_ntaps, low_pass), window.cc (kaiser, Izero).  Returns the prototype low-pass; ri/rd = reduced
// interpolation/decimation.
inline std::vector<float> resampler_taps(int interp, int decim, int &ri, int &rd)
{
  auto gcd = [](int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; };
  auto izero = [](double x) { double sum = 1, u = 1, halfx = x / 2.0; int n = 1; do { double t = halfx / (double)n; n += 1; t *= t; u *= t; sum += u; } while (u >= 1e-21 * sum); return sum; };
  const int d = gcd(interp, decim);
  ri = interp / d; rd = decim / d;
  const double fractional_bw = 0.4, beta = 7.0, halfband = 0.5, rate = (double)ri / (double)rd;
  double width, mid;
  if (rate >= 1.0) { width = halfband - fractional_bw; mid = halfband - width / 2.0; }
  else { width = rate * (halfband - fractional_bw); mid = rate * halfband - width / 2.0; }
  double gain = ri; const double fs = ri, atten = beta / 0.1102 + 8.7;
  int ntaps = (int)(atten * fs / (22.0 * width));
  if ((ntaps & 1) == 0) ntaps++;
  std::vector<float> w(ntaps), taps(ntaps);
  const double ibeta = 1.0 / izero(beta), inm1 = 1.0 / (double)(ntaps - 1);
  for (int i = 0; i < ntaps; i++) { const double t = 2 * i * inm1 - 1; w[i] = (float)(izero(beta * std::sqrt(1.0 - t * t)) * ibeta); }
  const int M = (ntaps - 1) / 2;
  const double fwT0 = 2 * M_PI * mid / fs;
  for (int n = -M; n <= M; n++) taps[n + M] = n == 0 ? (float)(fwT0 / M_PI * w[n + M]) : (float)(std::sin(n * fwT0) / (n * M_PI) * w[n + M]);
  double fmax = taps[M];
  for (int n = 1; n <= M; n++) fmax += 2 * taps[n + M];
  gain /= fmax;
  for (int i = 0; i < ntaps; i++) taps[i] = (float)(taps[i] * gain);
  return taps;
}
// polyphase branches as rational_resampler_base installs them: branch[i % ri][i / ri] = taps[i], zero padded; nt taps per branch
inline std::vector<float> resampler_branches(const std::vector<float> &taps, int ri, int &nt)
{
  nt = ((int)taps.size() + ri - 1) / ri;
  std::vector<float> br((size_t)ri * nt, 0.f);
  for (size_t i = 0; i < taps.size(); i++) br[(i % ri) * nt + i / ri] = taps[i];
  return br;
}

// GF(256) exp/log, poly 0x11d (reed_solomon.cc:48-89)
inline void gf_tables(uint8_t *exp512, uint8_t *log256)
{
  int reg = 1;
  log256[0] = 255;
  for (int i = 0; i < 255; i++) {
    exp512[i] = (uint8_t)reg; exp512[i + 255] = (uint8_t)reg; log256[reg] = (uint8_t)i;
    reg <<= 1; if (reg & 0x100) reg ^= 0x11d; reg &= 0xff;
  }
  exp512[510] = exp512[0]; exp512[511] = exp512[1];
}

// RS(255,239) generator g(x) = prod_{i<16} (x - alpha^i) (reed_solomon.cc:168-192) and the division table
// T[b] = b * (g_15 .. g_0): one row is XORed into the running remainder per input byte
// (R <- R*x + c mod g, with x^16 = sum g_k x^k).  Row layout: byte k = coefficient of x^k.
inline std::vector<uint8_t> rs_division_table()
{
  uint8_t ex[512], lg[256];
  gf_tables(ex, lg);
  auto mul = [&](int a, int b) -> int { return (a && b) ? ex[lg[a] + lg[b]] : 0; };
  int g[17] = {0}; g[0] = 1;                                   // g[k] = coefficient of x^k, built root by root
  for (int i = 0; i < 16; i++) {
    int root = ex[i];
    for (int k = 16; k > 0; k--) g[k] = g[k - 1] ^ mul(g[k], root);
    g[0] = mul(g[0], root);
  }
  std::vector<uint8_t> T(256 * 16);
  for (int b = 0; b < 256; b++) for (int k = 0; k < 16; k++) T[b * 16 + k] = (uint8_t)mul(b, g[k]);
  return T;
}

}  // namespace dvbt
