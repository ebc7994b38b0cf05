/* o_acq.c -- OFDM symbol acquisition (Van de Beek ML CP timing + fractional CFO).
 * TEST INFRASTRUCTURE (see dvbt_oracle.h).
 * Restates lib/ofdm_sym_acquisition_impl.cc: peak_detect_process :72-146, ml_sync :148-351,
 * ctor :379-449 (rho :390-391, detector constants :448), general_work :489-568.
 * Third-party arithmetic (VOLK generic kernels, gr_expj, gr::fast_atan2f) is not in the
 * reference tree; restated from their published definitions: |z|^2 = re*re+im*im,
 * a*conj(b), |z| = sqrtf(re*re+im*im), expj = (cosf, sinf); fast_atan2f (a table
 * approximation, abs error ~1e-4 rad) is replaced by atan2f -- PARITY UNPINNED, float
 * tolerance domain (the resulting common phase cancels in the equaliser). */
#include "dvbt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* A tracking window at the very left of a call (cp_start within 8 samples of N+cp-1, where the reference's asserts :152-153 would fire in a
 * debug build) makes the reference index in[], d_norm[] and d_corr[] up to 8 + ... samples BEFORE their start (:166-186): in[] then is the
 * stream's history in GNU Radio's circular buffer (the samples consumed by the previous calls), which is what is read here when the caller
 * says it is there (o_acq_work_hist); samples in front of the stream's first one read 0. */
#define O_ACQ_BACK 32
struct o_acq {
  int N, cp;
  float snr, rho;
  float avg, rise, fall, alpha;
  float phase; double phaseinc, nextphaseinc; int nextpos;
  int initial_acq, cp_start, to_consume, to_out, freq_count, freq_timeout;
  ocf *gamma, *derot, *corr; float *lambda, *norm, *phi; int *peak_pos;
  float last_eps;
  long long hist;            /* samples of the stream in memory in front of in[0] of the current call */
  long long avail;           /* samples of the stream in memory from in[0] on (o_acq_set_avail; 0 = unknown: everything asked for is read) */
};
#define O_ACQ_FWD 65536      /* how far beyond 2N+cp the tracking window may creep before the restatement stops following it */

o_acq *o_acq_new(const o_cfg *c, float snr_db)
{
  o_acq *a = calloc(1, sizeof *a);
  a->N = c->N; a->cp = c->cp;
  a->snr = (float)pow(10, snr_db / 10.0);                    /* :390 */
  a->rho = (float)(a->snr / (a->snr + 1.0));                 /* :391 */
  a->rise = 0.8f; a->fall = 0.9f; a->alpha = 0.9f; a->avg = 0;   /* :448 */
  a->freq_timeout = 0;                                       /* :386 */
  int W = 2 * a->N + a->cp;
  a->gamma = calloc(a->N, sizeof(ocf)); a->lambda = calloc(a->N, sizeof(float));
  a->phi = calloc(a->N, sizeof(float)); a->peak_pos = calloc(a->N, sizeof(int));
  a->derot = calloc(a->N + a->cp, sizeof(ocf));
  /* +16: in tracking the look-up window reaches cp_start + 8, and cp_start may sit at the very end of the 2N+cp samples the
   * reference forecasts (its own d_norm/d_corr are W long: it would write past them there); o_rx_run keeps 16 more samples visible */
  /* ... and beyond: the tracker follows its own peak, +-7 per call, and under noise it creeps past the window's end.  The reference then reads `in` beyond
   * what its forecast asked for (GNU Radio's buffer holds more) and WRITES beyond d_norm / d_corr (heap overflow: undefined).  Defined here as "the arrays are
   * long enough": the stream's own samples are read, zeros at and beyond its end (o_acq_set_avail); the product does the same (FrontParams.avail) */
  a->norm = (float *)calloc(W + 16 + O_ACQ_FWD + O_ACQ_BACK, sizeof(float)) + O_ACQ_BACK; a->corr = (ocf *)calloc(W + 16 + O_ACQ_FWD + O_ACQ_BACK, sizeof(ocf)) + O_ACQ_BACK;
  return a;
}

void o_acq_free(o_acq *a)
{
  if (!a) return;
  free(a->gamma); free(a->lambda); free(a->phi); free(a->peak_pos);
  free(a->derot); free(a->norm - O_ACQ_BACK); free(a->corr - O_ACQ_BACK); free(a);
}

/* :72-146 */
static int peak_detect(o_acq *a, const float *d, int n, int *peak_pos, int *peak_max)
{
  int state = 0, peak_index = 0, npk = 0, i = 0;
  float peak_val = -INFINITY;
  while (i < n) {
    if (state == 0) {
      if (d[i] > a->avg * a->rise) state = 1;
      else { a->avg = a->alpha * d[i] + (1 - a->alpha) * a->avg; i++; }
    } else {
      if (d[i] > peak_val) {
        peak_val = d[i]; peak_index = i;
        a->avg = a->alpha * d[i] + (1 - a->alpha) * a->avg; i++;
      } else if (d[i] > a->avg * a->fall) {
        a->avg = a->alpha * d[i] + (1 - a->alpha) * a->avg; i++;
      } else {
        peak_pos[npk++] = peak_index; state = 0; peak_val = -INFINITY;
      }
    }
  }
  if (npk) {
    float mx = d[peak_pos[0]]; int mi = 0;
    for (int k = 1; k < npk; k++) if (d[peak_pos[k]] > mx) { mx = d[peak_pos[k]]; mi = k; }
    *peak_max = mi;
  }
  return npk;
}

static void advance_phase(o_acq *a)
{
  a->phase = (float)((double)a->phase + a->phaseinc);
  while (a->phase > (float)M_PI) a->phase -= (float)(2.0 * M_PI);
  while (a->phase < (float)(-M_PI)) a->phase += (float)(2.0 * M_PI);
}

/* :148-351.  Samples the reference would read before in[0] are the stream's own (a->hist of them are in memory), 0 before the stream. */
static int ml_sync(o_acq *a, const ocf *in, int lookup_start, int lookup_stop)
{
  const int N = a->N, cp = a->cp;
  const int back = a->hist < O_ACQ_BACK ? (int)a->hist : O_ACQ_BACK;      /* how far in front of in[0] there are samples */
  int low = lookup_stop - (cp + N - 1);
  const long long avail = a->avail > 0 ? a->avail : (1ll << 62);
#define O_IN(i) ((long long)(i) < avail ? in[i] : (ocf)0.0f)
  for (int i = low < -back ? -back : low; i <= lookup_start; i++) {
    const ocf x = O_IN(i);
    float re = crealf(x), im = cimagf(x);
    a->norm[i] = re * re + im * im;
  }
  low = lookup_stop - cp - 1;
  for (int i = low; i <= lookup_start; i++) {
    if (i - N < -back) continue;
    a->corr[i - N] = O_IN(i) * conjf(O_IN(i - N));
  }
#undef O_IN
  for (int i = lookup_start - 1; i >= lookup_stop; i--) {
    int k = i - lookup_stop;
    float phi = 0.0f; ocf g = 0.0f;
    for (int j = 0; j < cp; j++) {
      int ci = i - j - N;
      if (ci < -back) continue;
      g += a->corr[ci];
      phi += a->norm[i - j] + a->norm[ci];
    }
    a->phi[k] = phi; a->gamma[k] = g;
  }
  int n = lookup_start - lookup_stop;
  float half_rho = (float)(a->rho / 2.0);
  for (int k = 0; k < n; k++) {
    float re = crealf(a->gamma[k]), im = cimagf(a->gamma[k]);
    a->lambda[k] = sqrtf(re * re + im * im);
    a->phi[k] = a->phi[k] * half_rho;
    a->lambda[k] = a->lambda[k] - a->phi[k];
  }
  int peak_max = 0;
  int npk = peak_detect(a, a->lambda, n, a->peak_pos, &peak_max);
  if (npk) {
    int peak = a->peak_pos[peak_max] + lookup_stop;
    a->cp_start = peak;
    ocf gp = a->gamma[a->peak_pos[peak_max]];
    float eps = atan2f(cimagf(gp), crealf(gp));
    a->last_eps = eps;
    double sensitivity = (double)(-1) / (double)N;
    for (int i = 0; i < cp + N; i++) {
      if (i == a->nextpos) a->phaseinc = a->nextphaseinc;
      advance_phase(a);
      a->derot[i] = cosf(a->phase) + I * sinf(a->phase);
    }
    a->nextphaseinc = sensitivity * eps;
    a->nextpos = peak - (cp + N);
    a->to_consume = cp + N; a->to_out = 1;
  } else {
    for (int i = 0; i < cp + N; i++) advance_phase(a);
    a->to_consume = cp + N; a->to_out = 0;
  }
  return npk;
}

/* :489-568 */
int o_acq_work(o_acq *a, const ocf *in, ocf *out, int *consumed, int *sync_start,
               int *cp_start, float *epsilon)
{ return o_acq_work_hist(a, in, 0, out, consumed, sync_start, cp_start, epsilon); }

void o_acq_set_avail(o_acq *a, long long avail) { a->avail = avail; }

int o_acq_work_hist(o_acq *a, const ocf *in, long long hist, ocf *out, int *consumed, int *sync_start,
                    int *cp_start, float *epsilon)
{
  a->hist = hist;
  const int N = a->N, cp = a->cp;
  *sync_start = 0;
  if (!a->initial_acq) {
    a->initial_acq = ml_sync(a, in, 2 * N + cp - 1, N + cp - 1);
    *sync_start = 1;                                          /* :507 */
  }
  if (a->initial_acq) {
    int found = ml_sync(a, in, a->cp_start + 8, a->cp_start - 8);
    if (found) {
      a->freq_count = 0;
      int low = a->cp_start - N + 1;
      const long long av = a->avail > 0 ? a->avail : (1ll << 62);
      for (int j = 0; j < N; j++) out[j] = a->derot[j] * ((long long)(low + j) < av ? in[low + j] : (ocf)0.0f);
    } else if (++a->freq_count > a->freq_timeout) {
      a->initial_acq = 0; a->freq_count = 0;
      a->to_consume = a->to_consume / 2;
    }
  }
  *consumed = a->to_consume;
  if (cp_start) *cp_start = a->cp_start;
  if (epsilon) *epsilon = a->last_eps;
  return a->to_out;
}
