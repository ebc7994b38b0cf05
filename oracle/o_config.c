/* o_config.c -- derived constants + pilot/TPS tables. TEST INFRASTRUCTURE (see dvbt_oracle.h).
 * Restates lib/dvbt_config.cc:94-250 and the carrier tables of
 * lib/reference_signals_impl.cc:54-126 (ETSI EN 300 744 tables 7 and 8). */
#include "dvbt_oracle.h"
#include <math.h>
#include <string.h>

/* EN 300 744 table 7, 2k mode (lib/reference_signals_impl.cc:54-62). */
static const int cpilot_2k[45] = {
  0, 48, 54, 87, 141, 156, 192, 201, 255, 279, 282, 333, 432, 450, 483, 525, 531, 618, 636,
  714, 759, 765, 780, 804, 873, 888, 918, 939, 942, 969, 984, 1050, 1101, 1107, 1110, 1137,
  1140, 1146, 1206, 1269, 1323, 1377, 1491, 1683, 1704 };
/* EN 300 744 table 8, 2k mode (lib/reference_signals_impl.cc:65-70). */
static const int tps_2k[17] = {
  34, 50, 209, 346, 413, 569, 595, 688, 790, 901, 1073, 1219, 1262, 1286, 1469, 1594, 1687 };

/* The 8k tables (lib/reference_signals_impl.cc:76-117) are the 2k tables repeated with
 * period 1704 (plus the final continual pilot at Kmax); built once here. */
static int cpilot_8k[177], tps_8k[68];
static int tables_ready = 0;

static void build_tables(void)
{
  if (tables_ready) return;
  int n = 0;
  for (int rep = 0; rep < 4; rep++)
    for (int i = 0; i < 44; i++) cpilot_8k[n++] = cpilot_2k[i] + 1704 * rep;
  cpilot_8k[n++] = 6816;
  n = 0;
  for (int rep = 0; rep < 4; rep++)
    for (int i = 0; i < 17; i++) tps_8k[n++] = tps_2k[i] + 1704 * rep;
  tables_ready = 1;
}

void o_cfg_init(o_cfg *c, int constellation, int hierarchy, int code_rate, int guard,
                int mode, int include_cell_id, int cell_id)
{
  build_tables();
  memset(c, 0, sizeof *c);
  c->constellation = constellation; c->hierarchy = hierarchy; c->code_rate = code_rate;
  c->guard = guard; c->mode = mode; c->include_cell_id = include_cell_id; c->cell_id = cell_id;

  /* dvbt_config.cc:105-125 */
  c->Kmin = 0;
  if (mode == O_T8k) { c->Kmax = 6816; c->N = 8192; c->payload = 6048; }
  else               { c->Kmax = 1704; c->N = 2048; c->payload = 1512; }
  c->zeros_left  = (int)ceil((c->N - (c->Kmax - c->Kmin + 1)) / 2.0);
  c->zeros_right = c->N - c->zeros_left - (c->Kmax - c->Kmin + 1);

  /* dvbt_config.cc:126-148 */
  c->step = 2;
  switch (constellation) {
    case O_QPSK:  c->csize = 4;  c->m = 2; break;
    case O_QAM64: c->csize = 64; c->m = 6; break;
    default:      c->csize = 16; c->m = 4; break;
  }
  /* dvbt_config.cc:150-192: the LP switch overwrites k/n; callers pass HP==LP */
  switch (code_rate) {
    case O_C2_3: c->k = 2; c->n = 3; break;
    case O_C3_4: c->k = 3; c->n = 4; break;
    case O_C5_6: c->k = 5; c->n = 6; break;
    case O_C7_8: c->k = 7; c->n = 8; break;
    default:     c->k = 1; c->n = 2; break;
  }
  /* dvbt_config.cc:194-211 */
  switch (guard) {
    case O_G1_16: c->cp = c->N / 16; break;
    case O_G1_8:  c->cp = c->N / 8;  break;
    case O_G1_4:  c->cp = c->N / 4;  break;
    default:      c->cp = c->N / 32; break;
  }
  /* dvbt_config.cc:213-225 */
  switch (hierarchy) {
    case O_ALPHA2: c->alpha = 2; break;
    case O_ALPHA4: c->alpha = 4; break;
    default:       c->alpha = 1; break;
  }
  /* dvbt_config.cc:229-249 (float d_norm = double expr) */
  double nrm;
  if (c->m == 2) nrm = 1.0 / sqrt(2);
  else if (c->m == 6) nrm = c->alpha == 1 ? 1.0 / sqrt(42) : c->alpha == 2 ? 1.0 / sqrt(60) : 1.0 / sqrt(108);
  else nrm = c->alpha == 1 ? 1.0 / sqrt(10) : c->alpha == 2 ? 1.0 / sqrt(20) : 1.0 / sqrt(52);
  c->norm = (float)nrm;

  if (mode == O_T8k) { c->cpilot = cpilot_8k; c->n_cpilot = 177; c->tps = tps_8k; c->n_tps = 68; c->n_spilot = 568; }
  else               { c->cpilot = cpilot_2k; c->n_cpilot = 45;  c->tps = tps_2k; c->n_tps = 17; c->n_spilot = 142; }
}

/* lib/reference_signals_impl.cc:334-345: w_k PRBS x^11+x^2+1, all-ones init */
void o_prbs_wk(const o_cfg *c, char *wk)
{
  unsigned reg = (1u << 11) - 1;
  for (int k = 0; k < c->Kmax - c->Kmin + 1; k++) {
    wk[k] = (char)(reg & 1);
    unsigned nb = ((reg >> 2) ^ reg) & 1;
    reg = (reg >> 1) | (nb << 10);
  }
}

/* BCH(67,53) LFSR shared by generate (:352-382) and verify (:385-425):
 * 60 leading zeros then bits s1..s53. returns the 14-bit register. */
static unsigned bch_reg(const unsigned char *tps68)
{
  unsigned reg = 0;
  for (int i = 0; i < 113; i++) {
    unsigned d = i < 60 ? 0 : tps68[1 + (i - 60)];
    unsigned fb = 1 & (d ^ reg);
    reg >>= 1;
    reg |= fb << 13;
    reg ^= (fb << 12) ^ (fb << 11) ^ (fb << 9) ^ (fb << 8) ^ (fb << 7) ^ (fb << 5) ^ (fb << 4);
  }
  return reg;
}

static void set_bits(unsigned char *t, int start, int stop, unsigned data)
{ /* reference_signals_impl.cc:852-859: LSB of data lands at index `start` */
  for (int i = start; i >= stop; i--) { t[i] = data & 1; data >>= 1; }
}

/* lib/reference_signals_impl.cc:883-916 */
void o_tps_format(const o_cfg *c, int frame_index, const char *wk, unsigned char *t)
{
  memset(t, 0, 68);
  set_bits(t, 0, 0, (unsigned)wk[0]);
  set_bits(t, 16, 1, (frame_index % 2) ? 0xca11 : 0x35ee);
  set_bits(t, 22, 17, c->include_cell_id ? 0x1f : 0x17);
  set_bits(t, 24, 23, (unsigned)frame_index);
  set_bits(t, 26, 25, (unsigned)c->constellation);
  set_bits(t, 29, 27, (unsigned)c->hierarchy);
  set_bits(t, 32, 30, (unsigned)c->code_rate);
  set_bits(t, 35, 33, (unsigned)c->code_rate);
  set_bits(t, 37, 36, (unsigned)c->guard);
  set_bits(t, 39, 38, (unsigned)c->mode);
  set_bits(t, 47, 40, (unsigned)c->cell_id);
  set_bits(t, 53, 48, 0);
  unsigned reg = bch_reg(t);
  for (int i = 0; i < 14; i++) t[54 + i] = 1 & (reg >> i);
}

/* lib/reference_signals_impl.cc:385-425: 0 = ok, -1 = parity mismatch */
int o_bch_check(const unsigned char *t)
{
  unsigned reg = bch_reg(t);
  for (int i = 0; i < 14; i++)
    if (t[54 + i] != (1 & (reg >> i))) return -1;
  return 0;
}
