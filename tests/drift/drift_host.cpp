// CPU check of gr_dvbt_amd/csrc/k_drift.hpp (the closed form of the reference's float phase accumulator) against the literal accumulator
// (ofdm_sym_acquisition_impl.cc:285-309 restated: float phase, double increment, one addition per sample, float wrap constants).
// usage: drift_host N cp nsym eps jitter seed  ->  prints: literal_wander sequential_residual parallel_residual parallel_entry_error
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include "../../gr_dvbt_amd/csrc/k_drift_math.hpp"
using namespace dvbt;

static double wrapd(double x) { x = fmod(x + M_PI, 2 * M_PI); if (x < 0) x += 2 * M_PI; return x - M_PI; }

int main(int argc, char **argv)
{
  if (argc < 7) return 2;
  const int N = atoi(argv[1]), cp = atoi(argv[2]), nsym = atoi(argv[3]), L = N + cp;
  const double eps0 = atof(argv[4]), jit = atof(argv[5]); unsigned seed = (unsigned)atoi(argv[6]);
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffffff) / 16777216.0 - 0.5; };
  std::vector<float> eps(nsym + 2); std::vector<int> sw(nsym);
  for (auto &e : eps) e = (float)(eps0 + jit * 3.4 * rnd());
  for (auto &w : sw) w = cp - 1 + (int)(20 * rnd());
  std::vector<double> incA(nsym), incB(nsym);
  for (int s = 0; s < nsym; s++) { incA[s] = s == 0 ? 0.0 : -(double)eps[s - 1] / N; incB[s] = -(double)eps[s] / N; }
  // ---- literal accumulator: deviation from the exact line at the steps 32 k + 17 of every call, and the phase at every call entry
  const int nb = N / 32;
  std::vector<double> lit((size_t)nsym * nb), lit_entry(nsym), ex_entry(nsym);
  {
    float ph = 0.f; double ex = 0.0;
    for (int s = 0; s < nsym; s++) {
      lit_entry[s] = ph; ex_entry[s] = ex;
      double inc = incA[s];
      for (int i = 0; i < L; i++) {
        if (i == sw[s]) inc = incB[s];
        ph = (float)((double)ph + inc); ex += inc;
        while (ph > (float)M_PI) ph -= (float)(2.0 * M_PI);
        while (ph < (float)(-M_PI)) ph += (float)(2.0 * M_PI);
        if (i < N && (i & 31) == 16) lit[(size_t)s * nb + (i >> 5)] = wrapd(wrapd((double)ph - ex) - wrapd(lit_entry[s] - ex_entry[s]));
      }
    }
  }
  auto delta_of = [&](int s, double ent, std::vector<double> &out) {   // what drift_table_kernel computes
    const double phsw = drift_advance(incA[s], ent, sw[s]);
    for (int k = 0; k < nb; k++) {
      const int n = 32 * k + 17;
      double phi, exact;
      if (n <= sw[s]) { phi = drift_advance(incA[s], ent, n); exact = n * incA[s]; }
      else { phi = drift_advance(incB[s], phsw, n - sw[s]); exact = sw[s] * incA[s] + (n - sw[s]) * incB[s]; }
      out[k] = (phi - ent) - exact;
    }
  };
  // ---- sequential use of the closed form (exact recurrence over the calls)
  double worst_lit = 0, worst_seq = 0, worst_par = 0, worst_ent = 0;
  std::vector<double> d(nb);
  {
    double ent = 0.0;
    for (int s = 0; s < nsym; s++) {
      delta_of(s, ent, d);
      for (int k = 0; k < nb; k++) { worst_seq = std::max(worst_seq, fabs(d[k] - lit[(size_t)s * nb + k])); worst_lit = std::max(worst_lit, fabs(lit[(size_t)s * nb + k])); }
      ent = drift_advance(incB[s], drift_advance(incA[s], ent, sw[s]), L - sw[s]);
    }
  }
  // ---- the parallel scheme of the kernels: tables per run, three rounds of (evaluate d_r at the previous iterate, prefix sum)
  {
    const bool neg = incB[0] < 0;
    std::vector<DriftTab> tab(nsym);
    for (int r = 0; r < nsym; r++) drift_build(incB[r], tab[r].q, tab[r].tc);
    std::vector<double> ex_run(nsym), S(nsym, 0.0), dd(nsym, 0.0);
    for (int r = 0; r < nsym; r++) ex_run[r] = ex_entry[r] + sw[r] * incA[r];
    const double A0 = drift_T(tab[0].q, tab[0].tc, neg, 0.0);
    auto phi_run = [&](int r, bool first) {
      if (r == 0) return 0.0;
      if (first) return ex_run[r];
      return drift_Tinv(tab[r - 1].q, tab[r - 1].tc, neg, A0 + ((double)r * L + sw[r] - sw[0]) - S[r - 1]);
    };
    for (int it = 0; it < 3; it++) {
      for (int r = 1; r < nsym; r++) { const double phi = phi_run(r, it == 0); dd[r] = drift_T(tab[r - 1].q, tab[r - 1].tc, neg, phi) - drift_T(tab[r].q, tab[r].tc, neg, phi); }
      double acc = 0.0; for (int r = 0; r < nsym; r++) { acc += dd[r]; S[r] = acc; }
    }
    for (int s = 0; s < nsym; s++) {
      double ent = 0.0;
      if (s > 0) { const double phr = phi_run(s - 1, false); ent = drift_Tinv(tab[s - 1].q, tab[s - 1].tc, neg, drift_T(tab[s - 1].q, tab[s - 1].tc, neg, phr) + (L - sw[s - 1])); }
      worst_ent = std::max(worst_ent, fabs(wrapd(ent - lit_entry[s])));
      delta_of(s, ent, d);
      for (int k = 0; k < nb; k++) worst_par = std::max(worst_par, fabs(d[k] - lit[(size_t)s * nb + k]));
    }
  }
  printf("%.3e %.3e %.3e %.3e\n", worst_lit, worst_seq, worst_par, worst_ent);
  return 0;
}
