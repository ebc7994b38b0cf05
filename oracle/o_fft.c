/* o_fft.c -- the DFT that sits between ofdm_sym_acquisition and demod_reference_signals.
 * TEST INFRASTRUCTURE (see dvbt_oracle.h).
 * The reference uses the stock gr::fft::fft_vcc block (FFTW3f, not vendored): forward,
 * rectangular window, shift=True, unnormalised (apps/dvbt_rx_demo*.grc block fft_vxx_0).
 * Published semantics restated here: X[k] = sum_n x[n] e^{-j2pi nk/N}; with shift the output
 * halves are swapped so out[b] = X[(b - N/2) mod N] (SURVEY C-2).  TX side: reverse, shift=True:
 * input halves swapped first, then x[n] = sum_k X[k] e^{+j2pi nk/N}, unnormalised.
 * Computed in double and rounded once to float (the "ideal" float result; FFTW3f differs
 * from it at the 1e-7 relative level, inside the stated tolerance of the A3 tap). */
#include "dvbt_oracle.h"
#include <math.h>
#include <stdlib.h>

typedef struct { int N; double *cs; int *rev; } plan_t;
static plan_t plans[4];

static plan_t *get_plan(int N)
{
  for (int i = 0; i < 4; i++) if (plans[i].N == N) return &plans[i];
  plan_t *p = NULL;
  for (int i = 0; i < 4; i++) if (plans[i].N == 0) { p = &plans[i]; break; }
  if (!p) p = &plans[0];
  p->N = N;
  p->cs = malloc(sizeof(double) * N);          /* cos/sin for k < N/2 */
  for (int k = 0; k < N / 2; k++) {
    p->cs[2 * k] = cos(2.0 * M_PI * k / N); p->cs[2 * k + 1] = sin(2.0 * M_PI * k / N);
  }
  p->rev = malloc(sizeof(int) * N);
  int bits = 0; while ((1 << bits) < N) bits++;
  for (int i = 0; i < N; i++) {
    int r = 0;
    for (int b = 0; b < bits; b++) if (i & (1 << b)) r |= 1 << (bits - 1 - b);
    p->rev[i] = r;
  }
  return p;
}

/* in-place radix-2 DIT on double pairs; sign=-1 forward, +1 inverse */
static void fft_core(plan_t *p, double *re, double *im, int sign)
{
  int N = p->N;
  for (int len = 2; len <= N; len <<= 1) {
    int half = len >> 1, stride = N / len;
    for (int s = 0; s < N; s += len)
      for (int j = 0; j < half; j++) {
        double wr = p->cs[2 * j * stride], wi = sign * p->cs[2 * j * stride + 1];
        int a = s + j, b = a + half;
        double tr = re[b] * wr - im[b] * wi, ti = re[b] * wi + im[b] * wr;
        re[b] = re[a] - tr; im[b] = im[a] - ti;
        re[a] += tr; im[a] += ti;
      }
  }
}

void o_fft_forward_shift(int N, const ocf *in, ocf *out)
{
  plan_t *p = get_plan(N);
  double *re = malloc(sizeof(double) * 2 * N), *im = re + N;
  for (int i = 0; i < N; i++) { re[p->rev[i]] = crealf(in[i]); im[p->rev[i]] = cimagf(in[i]); }
  fft_core(p, re, im, -1);
  for (int b = 0; b < N; b++) {
    int k = (b + N / 2) % N;
    out[b] = (float)re[k] + I * (float)im[k];
  }
  free(re);
}

void o_ifft_shift(int N, const ocf *in, ocf *out)
{
  plan_t *p = get_plan(N);
  double *re = malloc(sizeof(double) * 2 * N), *im = re + N;
  for (int k = 0; k < N; k++) {
    int b = (k + N / 2) % N;                   /* bin k of the IDFT reads shifted input b */
    re[p->rev[k]] = crealf(in[b]); im[p->rev[k]] = cimagf(in[b]);
  }
  fft_core(p, re, im, +1);
  for (int n = 0; n < N; n++) out[n] = (float)re[n] + I * (float)im[n];
  free(re);
}
