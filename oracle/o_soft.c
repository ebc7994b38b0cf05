/* o_soft.c -- an integer / float MODEL of the product's soft-decision kernels (gr_dvbt_amd/csrc/k_soft.hpp, k_soft4.hpp).
 *
 * TEST INFRASTRUCTURE, and NOT a restatement of the reference: gr-dvbt decodes hard decisions only (TODO.txt:25-26 lists "soft decision demapper / Viterbi";
 * the soft-metric table lib/d_metrics.c:34-133 is compiled out), so there is nothing of the reference's to restate.  This file states in plain C, independently of the
 * HIP code (no shared source, arrays and loops instead of lanes, DPP and LDS), WHAT those kernels are specified to compute:
 *   o_soft_demap    the 8-bit max-log likelihood ratio of every coded bit from the equalised carrier and its channel state (unit 8 per decision distance at the
 *                   symbol's mean channel power, weight capped at 4, clamp +-31), written to the bit's place behind both inner de-interleavers
 *                   (lib/symbol_inner_interleaver_impl.cc:197-209, lib/bit_inner_deinterleaver_impl.cc:120-184 applied to positions);
 *   o_soft_viterbi  the soft-input K = 7 decoder: depuncturing with erasures = 0 (lib/viterbi_decoder_impl.cc:241-256), correlation branch metrics, add-compare-
 *                   select with "the own offer wins a tie", chunks of B decoded bytes each decoded by its own decoder from all-zero metrics 256 steps early, traceback
 *                   in 8 segments per chunk, each from the best state recorded 24 decision groups behind the segment's end (smallest physical cell among equals).
 * tests/test_gpu_soft.py feeds both with the GPU's own taps and requires identical soft values and identical decoded bytes: a defect in the kernels' arithmetic,
 * tie rules, depuncturing or traceback shows up as a differing byte instead of as 0.3 dB nobody measures.  The trellis is kept in the kernels' "cell" coordinates
 * (cell c holds state rotl6(c, u mod 6) at relative step u) because the tie rules are defined there. */
#include "dvbt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SOFT_CLAMP 31
#define SOFT_UNIT 8.0f

/* where soft value x = v q + k of a symbol (bit k, MSB first, of word q behind both de-interleavers) comes from: carrier * v + label bit */
static void soft_dst_table(const o_cfg *c, const int *H, int odd, unsigned short *dst /* [P v]: carrier * v + e -> x */)
{
  static const int off[6] = {0, 63, 105, 42, 21, 84};
  const int P = c->payload, v = c->m;
  int *Hinv = malloc(sizeof(int) * P);
  for (int q = 0; q < P; q++) Hinv[H[q]] = q;
  for (int x = 0; x < P * v; x++) {
    const int q = x / v, k = x - q * v, b = q / 126, i = q - b * 126;
    const int e = ((v * i + k) % v) / (v / 2) + 2 * ((v * i + k) % (v / 2));
    int w = (i - off[e]) % 126; if (w < 0) w += 126;
    const int qq = b * 126 + w;
    dst[(size_t)(odd ? Hinv[qq] : H[qq]) * v + e] = (unsigned short)x;
  }
  free(Hinv);
}

/* eq, csi: nsym symbols of P carriers (the EQ and CSI taps of the symbol kernel); parity[s]: symbol index & 1 of symbol s; out: nsym x P m soft values */
void o_soft_demap(const o_cfg *c, const ocf *eq, const float *csi, const int *parity, size_t nsym, signed char *out)
{
  const int P = c->payload, m = c->m, pa = m / 2, nl = 1 << pa;
  ocf *points = malloc(sizeof(ocf) * c->csize); o_constellation(c, 1.0f, points);
  int *H = malloc(sizeof(int) * P); o_sym_H(c, H);
  unsigned short *dst[2];
  for (int o = 0; o < 2; o++) { dst[o] = malloc(sizeof(unsigned short) * (size_t)P * m); soft_dst_table(c, H, o, dst[o]); }
  float lev[2][8];
  for (int a = 0; a < 2; a++)
    for (int u = 0; u < nl; u++) {   /* level u of axis a: the point whose bits of that axis are u, the others 0 */
      int label = 0;
      for (int jj = 0; jj < pa; jj++) label |= ((u >> (pa - 1 - jj)) & 1) << (m - 1 - (2 * jj + a));
      lev[a][u] = a ? cimagf(points[label]) : crealf(points[label]);
    }
  const float step = 2.0f * c->norm, inv_step2 = 1.0f / (step * step);
  for (size_t s = 0; s < nsym; s++) {
    const ocf *e = eq + s * P; const float *cs = csi + s * P;
    /* mean channel power of the symbol, summed the way a 256-thread workgroup sums it: strided partial sums, then a binary tree */
    float red[256];
    for (int t = 0; t < 256; t++) { float acc = 0.f; for (int i = t; i < P; i += 256) acc += cs[i]; red[t] = acc; }
    for (int o = 128; o > 0; o >>= 1) for (int t = 0; t < o; t++) red[t] += red[t + o];
    const float inv_mean = (float)P / fmaxf(red[0], 1e-30f);
    signed char *so = out + s * (size_t)P * m;
    const unsigned short *d = dst[parity[s] & 1];
    for (int i = 0; i < P; i++) {
      float w = cs[i] * inv_mean; w = fminf(w, 4.0f);
      const float scale = inv_step2 * w * SOFT_UNIT;
      for (int a = 0; a < 2; a++) {
        const float z = a ? cimagf(e[i]) : crealf(e[i]);
        float d0[3] = {3.0e38f, 3.0e38f, 3.0e38f}, d1[3] = {3.0e38f, 3.0e38f, 3.0e38f};
        for (int u = 0; u < nl; u++) {
          const float dz = z - lev[a][u], dd = dz * dz;
          for (int jj = 0; jj < pa; jj++) { if ((u >> (pa - 1 - jj)) & 1) d1[jj] = fminf(d1[jj], dd); else d0[jj] = fminf(d0[jj], dd); }
        }
        for (int jj = 0; jj < pa; jj++) {
          float v = (d1[jj] - d0[jj]) * scale;
          v = fminf(fmaxf(rintf(v), (float)-SOFT_CLAMP), (float)SOFT_CLAMP);
          so[d[(size_t)i * m + 2 * jj + a]] = (signed char)(int)v;
        }
      }
    }
  }
  free(points); free(H); free(dst[0]); free(dst[1]);
}

/* ---- the decoder */
#define S4_WARM 256
#define S4_BLK 48
#define S4_BMAX 304
#define S4_LOOK 128
#define S4_G0 30
#define S4_NSEG 8
#define S4_PRE 24
#define S4_GRID 2048
#define S4_WAVES 4

/* chunk size and steps per decoder for a stream of total_out bytes (k_soft4.hpp::s4_plan) */
void o_soft_plan(long long total_out, int ntb, int *B_out, int *nsteps_out)
{
  const long long per_round = 4ll * S4_GRID * S4_WAVES;
  long long rounds = (total_out + per_round * 256 / 2) / (per_round * 256); if (rounds < 1) rounds = 1;
  long long B;
  for (;;) {
    B = (total_out + rounds * per_round - 1) / (rounds * per_round);
    if (B < 64) B = 64;
    if (B <= S4_BMAX) break;
    rounds++;
  }
  const int look = 8 * ntb > S4_LOOK ? 8 * ntb : S4_LOOK;
  *B_out = (int)B; *nsteps_out = ((S4_WARM + 8 * (int)B + look + S4_BLK - 1) / S4_BLK) * S4_BLK;
}

static int rotl6(int c, int p) { p %= 6; return ((c << p) | (c >> (6 - p))) & 63; }
/* logical lane-in-row a3 a2 a1 a0 <-> physical (the half-mirror exchange of a2 is a DPP row_half_mirror) */
static int phys4(int a) { const int a2 = (a >> 2) & 1; return (a & 8) | (a2 << 2) | ((a & 3) ^ (a2 ? 3 : 0)); }
static int log4(int p) { const int q = p & 7, a2 = (q >> 2) & 1; return (p & 8) | (a2 << 2) | ((q ^ (a2 ? 7 : 0)) & 3); }
static int z_of(int c) { return phys4(c & 15) | (((c >> 4) & 1) << 6) | (((c >> 5) & 1) << 7); }
static int c_of(int z) { return (((z >> 7) & 1) << 5) | (((z >> 6) & 1) << 4) | log4(z & 15); }
static int decoded_bit(int z, int p1)
{
  switch (p1) { case 0: return (z ^ (z >> 2)) & 1; case 1: return (z >> 7) & 1; case 2: return (z >> 6) & 1; case 3: return (z >> 3) & 1; case 4: return (z >> 2) & 1; default: return ((z >> 1) ^ (z >> 2)) & 1; }
}

/* soft: n_soft values in the decoder's input order; total_steps trellis steps exist; B, nsteps from o_soft_plan of the LAUNCH's stream bound; out: total_steps / 8 - ntb bytes.
 * Returns the bytes written. */
long long o_soft_viterbi(const o_cfg *c, const signed char *soft, long long n_soft, long long total_steps, int B, int nsteps, unsigned char *out)
{
  static const int FLIP[6] = {0x80, 0x40, 8, 7, 2, 1};
  int plen; const unsigned char *punct = o_vit_puncture(c->code_rate, &plen);
  int prefix[17]; prefix[0] = 0; for (int i = 0; i < plen; i++) prefix[i + 1] = prefix[i] + punct[i];
  const int n = prefix[plen], ntb = o_vit_ntraceback(c->code_rate);
  const long long total_out = total_steps / 8 - ntb;
  if (total_out <= 0) return 0;
  const int ngrp = nsteps / 8, nblk = nsteps / S4_BLK;
  unsigned char *dec = malloc((size_t)nsteps * 64);                 /* decision of every cell and step: 1 = the survivor came from the butterfly partner */
  int *bests = malloc(sizeof(int) * nblk);
  int cls[6][64];
  for (int p = 0; p < 6; p++) for (int cc = 0; cc < 64; cc++) {
    const int i = rotl6(cc, p) & 31, c0 = ((i >> 2) ^ (i >> 1) ^ i) & 1, c1 = ((i >> 4) ^ (i >> 2) ^ (i >> 1)) & 1;
    cls[p][cc] = c0 | (c1 << 1);
  }
  for (long long b0 = 0; b0 < total_out; b0 += B) {
    const long long t0 = 8 * b0 - S4_WARM, tbase = t0 > 0 ? t0 : 0;
    const int skip = (int)(tbase - t0);
    const int phb = (int)((2 * tbase) % plen); const long long rb = (2 * tbase) / plen * n;
    int v[64]; memset(v, 0, sizeof v);
    for (int u = 0; u < nsteps; u++) {
      const int ir = u - skip;
      int sx = 0, sy = 0;
      if (ir >= 0 && tbase + ir < total_steps) {
        const int x = phb + 2 * ir, ph = x % plen; const long long r = rb + (long long)(x / plen) * n;
        if (punct[ph]) { const long long rr = r + prefix[ph]; if (rr < n_soft) sx = soft[rr]; }
        if (punct[ph + 1]) { const long long rr = r + prefix[ph + 1]; if (rr < n_soft) sy = soft[rr]; }
      }
      const int p = u % 6, A = sx + sy, Bm = sy - sx, dlt[4] = {A, Bm, -Bm, -A};
      int X[64], Y[64];
      for (int cc = 0; cc < 64; cc++) { const int d = dlt[cls[p][cc]]; X[cc] = v[cc] + d; Y[cc] = v[cc] - d; }
      for (int cc = 0; cc < 64; cc++) { const int yp = Y[cc ^ (1 << (5 - p))]; dec[(size_t)u * 64 + cc] = yp > X[cc]; v[cc] = yp > X[cc] ? yp : X[cc]; }
      if ((u + 1) % S4_BLK == 0) {
        int mx = v[0]; for (int cc = 1; cc < 64; cc++) if (v[cc] > mx) mx = v[cc];
        int bz = 0x100;
        for (int cc = 0; cc < 64; cc++) { v[cc] -= mx; if (v[cc] == 0 && z_of(cc) < bz) bz = z_of(cc); }
        bests[(u + 1) / S4_BLK - 1] = bz;
      }
    }
    const long long ob1 = b0 + B < total_out ? b0 + B : total_out;
    const int S = ((S4_WARM / 8 + B - S4_G0 + S4_NSEG * 6 - 1) / (S4_NSEG * 6)) * 6;
    for (int sg = 0; sg < S4_NSEG; sg++) {
      const int lo = S4_G0 + sg * S, hi = lo + S;
      int top = S4_G0 + (sg + 1) * S + S4_PRE; if (top > ngrp) top = ngrp;
      int z = bests[top / 6 - 1];
      for (int g = top - 1; g >= lo && g >= S4_G0; g--) {
        unsigned byte = 0;
        for (int jj = 7; jj >= 0; jj--) {
          const int J = 8 * g + jj, p = J % 6, p1 = (J + 1) % 6;
          byte |= (unsigned)decoded_bit(z, p1) << (7 - jj);
          if (dec[(size_t)J * 64 + c_of(z)]) z ^= FLIP[p];
        }
        const long long ob = b0 + g - S4_WARM / 8;
        if (g < hi && ob >= b0 && ob < ob1) out[ob] = (unsigned char)byte;
      }
    }
  }
  free(dec); free(bests);
  return total_out;
}
