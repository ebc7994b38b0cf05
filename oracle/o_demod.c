/* o_demod.c -- pilot/TPS engine (RX half of pilot_gen) + demod_reference_signals block logic.
 * TEST INFRASTRUCTURE (see dvbt_oracle.h).
 * Restates lib/reference_signals_impl.cc: ctor :131-332 (known_phase_diff :224-228),
 * process_cpilot_data :715-744, compute_oneshot_csft :747-790, frequency_correction :793-819,
 * process_spilot_data :536-689, process_tps_data :919-1032, process_payload_data :1065-1124,
 * parse_input :1189-1248; and lib/demod_reference_signals_impl.cc:73-77,97-150.
 * PARITY UNPINNED by reference execution (needs GNU Radio headers); float-tolerance tap. */
#include "dvbt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct o_demod {
  o_cfg c;
  char *wk;
  float *known_phase_diff;
  ocf *gain, *derot, *prev_tps;
  int *chanestim, *payload_carriers;
  unsigned char fifo[68];
  int symbol_index, symbol_index_known, frame_index;
  int freq_offset, mod_index, prev_mod_index, equalizer_ready;
  float carrier_freq_correction, sampling_freq_correction;
  int d_init, fi_start;
};

static inline ocf pilot_value(const char *wk, int k)
{ /* get_spilot_value :480-484 / get_cpilot_value :699-703: +-4/3 */
  return (float)(4 * 2 * (0.5 - wk[k]) / 3) + 0.0f * I;
}

o_demod *o_demod_new(const o_cfg *c)
{
  o_demod *d = calloc(1, sizeof *d);
  d->c = *c;
  int K = c->Kmax - c->Kmin + 1;
  d->wk = malloc(K); o_prbs_wk(c, d->wk);
  d->known_phase_diff = malloc(sizeof(float) * (c->n_cpilot - 1));
  for (int i = 0; i < c->n_cpilot - 1; i++) {
    ocf df = pilot_value(d->wk, c->cpilot[i + 1]) - pilot_value(d->wk, c->cpilot[i]);
    d->known_phase_diff[i] = crealf(df) * crealf(df) + cimagf(df) * cimagf(df);
  }
  d->gain = calloc(K, sizeof(ocf)); d->derot = calloc(c->N, sizeof(ocf));
  d->prev_tps = calloc(c->n_tps, sizeof(ocf));
  d->chanestim = calloc(K, sizeof(int)); d->payload_carriers = calloc(K, sizeof(int));
  /* demod_reference_signals_impl.cc:73-77 */
  d->fi_start = (c->constellation == O_QAM64 && c->mode == O_T8k) ? 2 : 3;
  return d;
}

void o_demod_free(o_demod *d)
{
  if (!d) return;
  free(d->wk); free(d->known_phase_diff); free(d->gain); free(d->derot);
  free(d->prev_tps); free(d->chanestim); free(d->payload_carriers); free(d);
}

static inline float cnorm(ocf z) { return crealf(z) * crealf(z) + cimagf(z) * cimagf(z); }

/* list of scattered pilots for pattern s, ascending (the k-loop of :549-563) */
static int spilot_list(const o_cfg *c, int s, int *out)
{
  int size = c->n_spilot + (s == 0 ? 1 : 0), n = 0;       /* advance_spilot :492-504 */
  for (int p = 0; p < size; p++) {
    int k = 3 * (s % 4) + 12 * p;
    if (k > c->Kmax) break;
    out[n++] = k;
  }
  return n;
}

static int parse_input(o_demod *d, const ocf *in, ocf *out, int *symbol_index, int *frame_index, int *info)
{
  const o_cfg *c = &d->c;
  const int N = c->N, zl = c->zeros_left, K = c->Kmax - c->Kmin + 1;

  /* --- process_cpilot_data :715-744 : integer CFO */
  {
    float max = 0, sum; int start = 0;
    for (int i = zl - 8; i < zl + 8; i++) {
      sum = 0;
      for (int j = 0; j < c->n_cpilot - 1; j++) {
        float ph = cnorm(in[i + c->cpilot[j + 1]] - in[i + c->cpilot[j]]);
        sum += d->known_phase_diff[j] * ph;
      }
      if (sum > max) { max = sum; start = i; }
    }
    d->freq_offset = start - zl;
  }
  /* --- compute_oneshot_csft :747-790 (reads the NEXT item at in+N) */
  {
    ocf left = 0.0f, right = 0.0f;
    int half = (c->n_cpilot - 1) / 2;
    float carrier_coeff = (float)(1.0 / (2 * M_PI * (1 + (float)c->cp / (float)N) * 2));
    float sampling_coeff = (float)(1.0 / (2 * M_PI * ((1 + (float)c->cp / (float)N) * ((float)c->n_cpilot / 2.0))));
    for (int j = 0; j < half; j++)
      left += in[d->freq_offset + zl + c->cpilot[j]] * conjf(in[d->freq_offset + N + zl + c->cpilot[j]]);
    for (int j = half + 1; j < c->n_cpilot; j++)
      right += in[d->freq_offset + zl + c->cpilot[j]] * conjf(in[d->freq_offset + N + zl + c->cpilot[j]]);
    float la = atan2f(cimagf(left), crealf(left)), ra = atan2f(cimagf(right), crealf(right));
    d->carrier_freq_correction = (ra + la) * carrier_coeff;
    d->sampling_freq_correction = (ra - la) * sampling_coeff;
  }
  /* --- frequency_correction :793-819: one constant phasor, input shifted by freq_offset */
  {
    float correction = (float)d->freq_offset + d->carrier_freq_correction;
    float ph = (float)(-2 * M_PI * correction * (N + c->cp) / N * 1);
    ocf cph = cosf(ph) + I * sinf(ph);
    for (int k = 0; k < N; k++) {
      int s = k + d->freq_offset;
      d->derot[k] = (s >= 0 && s < 2 * N) ? cph * in[s] : 0.0f;   /* s<0 only in the guard band */
    }
  }
  const ocf *x = d->derot;
  /* --- process_spilot_data :536-689 */
  int diff;
  {
    float max = 0, sum = 0;
    int *lst = d->chanestim;
    for (int s = 0; s < 4; s++) {
      spilot_list(c, s, lst);
      ocf acc = 0.0f;
      for (int j = 0; j < 10; j++) acc += pilot_value(d->wk, lst[j]) * conjf(x[zl + lst[j]]);
      sum = cnorm(acc);
      if (sum > max) { max = sum; d->mod_index = s; }
    }
    /* merged scattered+continual list in the order of the k-loop :597-614 (duplicates kept) */
    int n = 0, sp = 0, cpi = 0;
    int sp_size = c->n_spilot + (d->mod_index == 0 ? 1 : 0);
    for (int k = 0; k < K; k++) {
      if (k == 3 * (d->mod_index % 4) + 12 * sp) { lst[n++] = k; sp = (sp + 1) % sp_size; }
      if (k == c->cpilot[cpi]) { lst[n++] = k; cpi = (cpi + 1) % c->n_cpilot; }
    }
    for (int i = 0, startk = lst[0]; i < n; i++) {
      int k = lst[i];
      d->gain[k] = pilot_value(d->wk, k) / x[k + zl];                       /* set_channel_gain :486-490 */
      ocf tg = (d->gain[k] - d->gain[startk]) / (11.0f + 0.0f * I);         /* the constant 11 (:625) */
      for (int j = 1; j < k - startk; j++)
        d->gain[startk + j] = d->gain[startk] + tg * ((float)j + 0.0f * I);
      startk = k;
    }
    d->equalizer_ready = 1;
    diff = (d->mod_index - d->prev_mod_index + 4) % 4;
    d->prev_mod_index = d->mod_index;
  }
  /* --- parse_input :1228-1233 */
  d->symbol_index = (d->symbol_index + diff) % 68;
  *symbol_index = d->symbol_index;
  *frame_index = d->frame_index;

  /* --- process_tps_data :919-1032 */
  int end_frame = 0;
  {
    int maj = 0;
    int use = (!d->symbol_index_known || d->symbol_index != 0);
    for (int k = 0; k < c->n_tps; k++) {
      ocf val = x[zl + c->tps[k]] * d->gain[c->tps[k]];
      if (use) {
        ocf pd = val * conjf(d->prev_tps[k]);
        if (crealf(pd) >= 0.0f) maj++; else maj--;
      }
      d->prev_tps[k] = val;
    }
    if (info) info[3] = maj;
    for (int i = 0; i < diff; i++) {
      memmove(d->fifo, d->fifo + 1, 67);
      d->fifo[67] = use ? (maj >= 0 ? 0 : 1) : 0;
    }
    static const unsigned char sync_even[16] = { 0,0,1,1,0,1,0,1,1,1,1,0,1,1,1,0 };
    static const unsigned char sync_odd[16]  = { 1,1,0,0,1,0,1,0,0,0,0,1,0,0,0,1 };
    int m_even = memcmp(d->fifo + 1, sync_even, 15) == 0;    /* only 15 of 16 bits compared (B-11) */
    int m_odd  = !m_even && memcmp(d->fifo + 1, sync_odd, 15) == 0;
    if (m_even || m_odd) {
      if (o_bch_check(d->fifo) == 0) {
        d->frame_index = (d->fifo[23] << 1) | d->fifo[24];
        d->symbol_index_known = 1; end_frame = 1;
      } else { d->symbol_index_known = 0; end_frame = 0; }
      memset(d->fifo, 0, 68);
    }
  }
  if (end_frame) d->symbol_index = 67;                         /* :1240-1241 */

  /* --- process_payload_data :1065-1124 */
  {
    int np = 0, sp = 0, cpi = 0, tpi = 0;
    int sp_size = c->n_spilot + (d->mod_index == 0 ? 1 : 0);
    for (int k = 0; k < K; k++) {
      int is_payload = 1;
      if (k == 3 * (d->mod_index % 4) + 12 * sp) { sp = (sp + 1) % sp_size; is_payload = 0; }
      if (k == c->cpilot[cpi]) { cpi = (cpi + 1) % c->n_cpilot; is_payload = 0; }
      if (k == c->tps[tpi]) { tpi = (tpi + 1) % c->n_tps; is_payload = 0; }
      if (is_payload) d->payload_carriers[np++] = k;
    }
    for (int i = 0; i < np; i++) out[i] = x[zl + d->payload_carriers[i]] * d->gain[d->payload_carriers[i]];
    if (info) info[4] = np;
  }
  if (info) { info[0] = d->freq_offset; info[1] = d->mod_index; info[2] = end_frame;
              info[5] = d->symbol_index; info[6] = d->frame_index; info[7] = d->symbol_index_known; }
  return 1;
}

/* lib/demod_reference_signals_impl.cc:97-150 in the one-item-per-call regime (SURVEY B-7) */
int o_demod_work(o_demod *d, const ocf *in, ocf *out, int sync_start_tag,
                 int *superframe_start, int *symbol_index, int *info)
{
  int si = 0, fi = 0;
  parse_input(d, in, out, &si, &fi, info);
  *superframe_start = 0;
  *symbol_index = si;
  if (sync_start_tag) d->d_init = 0;
  if (d->d_init == 0) {
    if ((si % 68) == 0 && (fi % 4) == d->fi_start) { d->d_init = 1; *superframe_start = 1; }
    else return 0;
  }
  return 1;
}
