/* o_resample.c -- front of the RX flowgraph: rational_resampler_ccc(64, 70) + multiply_const.
 * TEST INFRASTRUCTURE (see dvbt_oracle.h).
 *
 * The reference has no source for this stage: apps/dvbt_rx_demo*.grc instantiates GNU Radio's stock blocks
 *   rational_resampler_xxx_0: type ccc, interp 64, decim 70, taps "" (none), fbw 0 (none)
 *   blocks_multiply_const_vxx_0: complex, const 0.0022097087 (2k) / 0.00055242272 (8k)
 * gr-filter / gr-fft are third-party dependencies that are absent from /root/reference and not version-pinned
 * (CMakeLists.txt:88 asks for GNU Radio >= 3.7.2.1).  PARITY UNPINNED: this file restates the published
 * algorithm of the GNU Radio 3.7 series:
 *   gr-filter/python/filter/rational_resampler.py : interp/decim reduced by their gcd when no taps are given,
 *       fractional_bw 0.4, design_filter() = firdes.low_pass(interp, interp, mid, width, WIN_KAISER, beta 7.0)
 *   gr-filter/lib/firdes.cc : compute_ntaps() (a*fs/(22*width), made odd; a = beta/0.1102 + 8.7 for Kaiser),
 *       low_pass() (windowed sinc in float, normalised to `gain` at DC)
 *   gr-fft/lib/window.cc : kaiser() with the Izero series (epsilon 1e-21)
 *   gr-filter/lib/rational_resampler_base_XXX_impl.cc.t : taps dealt round-robin to `interp` FIR branches (zero
 *       padded to a multiple), general_work(): out = branch[ctr](in); ctr += decim; while (ctr >= interp) { ctr -= interp; in++ }
 *   gr-filter/lib/fir_filter.cc : y = sum_k taps[k] * x[n + ntaps-1-k], history ntaps-1 (zeros before the stream)
 * The dot products of the reference run in VOLK (summation order unpinned); here they are plain sequential float sums,
 * so this tap is a float-tolerance tap like the FFT. */
#include "dvbt_oracle.h"
#include <math.h>
#include <complex.h>
#include <stdlib.h>
#include <string.h>

static double izero(double x)
{
  double sum = 1, u = 1, halfx = x / 2.0; int n = 1;
  do { double t = halfx / (double)n; n += 1; t *= t; u *= t; sum += u; } while (u >= 1e-21 * sum);
  return sum;
}

static int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

/* design for rational_resampler(interp, decim, taps=None, fractional_bw=None).  Returns the tap count; *ri, *rd = the
 * reduced interpolation / decimation.  taps may be NULL to query the size. */
int o_resampler_design(int interp, int decim, int *ri, int *rd, float *taps, int cap)
{
  int d = gcd_i(interp, decim);
  interp /= d; decim /= d;
  *ri = interp; *rd = decim;
  const double fractional_bw = 0.4, beta = 7.0, halfband = 0.5;
  double rate = (double)interp / (double)decim, trans_width, mid;
  if (rate >= 1.0) { trans_width = halfband - fractional_bw; mid = halfband - trans_width / 2.0; }
  else { trans_width = rate * (halfband - fractional_bw); mid = rate * halfband - trans_width / 2.0; }
  double gain = interp, fs = interp;
  double a = beta / 0.1102 + 8.7;
  int ntaps = (int)(a * fs / (22.0 * trans_width));
  if ((ntaps & 1) == 0) ntaps++;
  if (!taps) return ntaps;
  if (cap < ntaps) return -1;
  float *w = malloc(sizeof(float) * ntaps);
  double ibeta = 1.0 / izero(beta), inm1 = 1.0 / (double)(ntaps - 1);
  for (int i = 0; i < ntaps; i++) { double t = 2 * i * inm1 - 1; w[i] = (float)(izero(beta * sqrt(1.0 - t * t)) * ibeta); }
  int M = (ntaps - 1) / 2;
  double fwT0 = 2 * M_PI * mid / fs;
  for (int n = -M; n <= M; n++) {
    if (n == 0) taps[n + M] = (float)(fwT0 / M_PI * w[n + M]);
    else taps[n + M] = (float)(sin(n * fwT0) / (n * M_PI) * w[n + M]);
  }
  double fmax = taps[0 + M];
  for (int n = 1; n <= M; n++) fmax += 2 * taps[n + M];
  gain /= fmax;
  for (int i = 0; i < ntaps; i++) taps[i] = (float)(taps[i] * gain);
  free(w);
  return ntaps;
}

/* number of outputs produced from n inputs by a stream that starts at phase 0 */
size_t o_resampler_nout(int interp, int decim, size_t nin)
{
  int ri, rd; o_resampler_design(interp, decim, &ri, &rd, NULL, 0);
  /* output m consumes inputs up to floor(m*rd/ri); it exists while that index < nin */
  return (size_t)(((unsigned long long)nin * ri + rd - 1) / rd);
}

/* whole-stream resample + scale: out[m] = scale * sum_j branch[ctr][j] * in[n + (nt-1) - j - (nt-1)]  (zeros before the stream) */
size_t o_resample_scale(int interp, int decim, float scale, const ocf *in, size_t nin, ocf *out, size_t cap)
{
  int ri, rd;
  int ntaps = o_resampler_design(interp, decim, &ri, &rd, NULL, 0);
  float *taps = malloc(sizeof(float) * ntaps);
  o_resampler_design(interp, decim, &ri, &rd, taps, ntaps);
  int nt = (ntaps + ri - 1) / ri;                       /* taps per branch after zero padding */
  float *br = calloc((size_t)ri * nt, sizeof(float));
  for (int i = 0; i < ntaps; i++) br[(size_t)(i % ri) * nt + i / ri] = taps[i];
  size_t m = 0; long long n = 0; int ctr = 0;
  while (m < cap && (size_t)n < nin) {
    /* fir_filter: y = sum_k t[k] * x[n + nt-1-k] with x shifted by the history of nt-1 zeros: x[n + nt-1-k] = in[n - k] */
    float ar = 0.f, ai = 0.f;
    const float *t = br + (size_t)ctr * nt;
    for (int k = nt - 1; k >= 0; k--) {                 /* reversed taps, ascending input index (fir_filter.cc) */
      long long idx = n - k;
      if (idx < 0) continue;
      /* complex taps with zero imaginary part (type ccc) */
      ar += crealf(in[idx]) * t[k]; ai += cimagf(in[idx]) * t[k];
    }
    out[m] = ar * scale + I * (ai * scale);              /* multiply_const (complex const with zero imaginary part) */
    m++;
    ctr += rd;
    while (ctr >= ri) { ctr -= ri; n++; }
  }
  free(br); free(taps);
  return m;
}
