/* o_rs.c -- GF(2^8) RS(255,239) shortened (204,188). TEST INFRASTRUCTURE (see dvbt_oracle.h).
 * Restates lib/reed_solomon.cc (gf_init :48-89, rs_init :150-203, rs_encode :216-244,
 * rs_decode :246-489) and the block body lib/reed_solomon_dec_impl.cc:77-116.
 * Parameters fixed to the values every demo flowgraph uses: p=2 m=8 poly=0x11d n=255 k=239 t=8 s=51. */
#include "dvbt_oracle.h"
#include <string.h>

#define RS_N 255
#define RS_K 239
#define RS_T 8
#define RS_S 51

static inline int g_exp(const o_rs *rs, int a) { return rs->exp[a % RS_N]; }            /* :98-102 */
static inline int g_mul(const o_rs *rs, int a, int b)                                    /* :118-125 */
{ return (a == 0 || b == 0) ? 0 : g_exp(rs, rs->log[a] + rs->log[b]); }
static inline int g_div(const o_rs *rs, int a, int b)                                    /* :127-134 */
{ return (a == 0 || b == 0) ? 0 : g_exp(rs, RS_N + rs->log[a] - rs->log[b]); }
static inline int g_pow(const o_rs *rs, int a, int p)                                    /* :136-143 */
{ return a == 0 ? 0 : g_exp(rs, RS_N + rs->log[a] + p); }

void o_rs_init(o_rs *rs)
{
  /* gf_init :48-89 */
  int reg = 1;
  rs->exp[255] = 0; rs->log[0] = 255;
  for (int i = 0; i < 255; i++) {
    rs->exp[i] = (unsigned char)reg; rs->log[reg] = (unsigned char)i;
    reg <<= 1;
    if (reg & 0x100) reg ^= 0x11d;
    reg &= 0xff;
  }
  /* rs_init :150-203 with lambda = p = 2 */
  rs->l[0] = 1;
  for (int i = 1; i <= RS_N; i++) rs->l[i] = (unsigned char)g_mul(rs, rs->l[i - 1], 2);
  memset(rs->g, 0, sizeof rs->g);
  rs->g[0] = 1;
  for (int i = 1; i <= 2 * RS_T; i++) {
    for (int j = i; j > 0; j--) {
      if (rs->g[j] != 0) rs->g[j] = rs->g[j - 1] ^ g_mul(rs, rs->g[j], rs->l[i - 1]);
      else rs->g[j] = rs->g[j - 1];
    }
    rs->g[0] = (unsigned char)g_mul(rs, rs->g[0], rs->l[i - 1]);
  }
}

/* rs_encode :216-244 */
void o_rs_encode(const o_rs *rs, const unsigned char *data, unsigned char *parity)
{
  memset(parity, 0, 2 * RS_T);
  for (int i = 0; i < RS_K; i++) {
    int fb = data[i] ^ parity[0];
    if (fb)
      for (int j = 1; j < 2 * RS_T; j++)
        if (rs->g[2 * RS_T - j]) parity[j] ^= g_mul(rs, fb, rs->g[2 * RS_T - j]);
    memmove(parity, parity + 1, 2 * RS_T - 1);
    parity[2 * RS_T - 1] = fb ? (unsigned char)g_mul(rs, fb, rs->g[0]) : 0;
  }
}

/* rs_decode :246-489 (errors only: the block passes eras=NULL, no_eras=0) */
int o_rs_decode(const o_rs *rs, unsigned char *data, int compat)
{
  unsigned char sigma[2 * RS_T + 1], b[2 * RS_T + 1], T[2 * RS_T + 1], reg[2 * RS_T + 1];
  unsigned char root[2 * RS_T + 1], loc[2 * RS_T + 1], omega[2 * RS_T + 1], syn[2 * RS_T];

  memset(sigma, 0, sizeof sigma); sigma[0] = 1;

  /* syndromes :281-288 (Horner, multiply by alpha^i) */
  for (int j = 0; j < 2 * RS_T; j++) syn[j] = data[0];
  for (int j = 1; j < RS_N; j++)
    for (int i = 0; i < 2 * RS_T; i++) syn[i] = data[j] ^ g_pow(rs, syn[i], i);
  int syn_error = 0;
  for (int i = 0; i < 2 * RS_T; i++) syn_error |= syn[i];
  if (!syn_error) return 0;                                   /* :299-305 */

  /* Berlekamp-Massey :315-354 */
  int r = 0, el = 0;
  memcpy(b, sigma, sizeof sigma);
  while (++r <= 2 * RS_T) {
    int discr = 0;
    for (int i = 0; i < r; i++) discr ^= g_mul(rs, sigma[i], syn[r - i - 1]);
    if (discr == 0) {
      memmove(&b[1], b, 2 * RS_T); b[0] = 0;
    } else {
      T[0] = sigma[0];
      for (int i = 0; i < 2 * RS_T; i++) T[i + 1] = sigma[i + 1] ^ g_mul(rs, discr, b[i]);
      if (2 * el <= r - 1) {
        el = r - el;
        for (int i = 0; i <= 2 * RS_T; i++) b[i] = (unsigned char)g_div(rs, sigma[i], discr);
      } else {
        memmove(&b[1], b, 2 * RS_T); b[0] = 0;
      }
      memcpy(sigma, T, sizeof sigma);
    }
  }
  int deg_sigma = 0;
  for (int i = 0; i < 2 * RS_T + 1; i++) if (sigma[i]) deg_sigma = i;

  /* Chien :376-403 */
  int no_roots = 0;
  memcpy(&reg[1], &sigma[1], 2 * RS_T);
  for (int i = 1; i <= RS_N; i++) {
    int q = 1;
    for (int j = deg_sigma; j > 0; j--) { reg[j] = (unsigned char)g_pow(rs, reg[j], j); q ^= reg[j]; }
    if (q != 0) continue;
    root[no_roots] = (unsigned char)i; loc[no_roots] = (unsigned char)(i - 1);
    if (++no_roots == deg_sigma) break;
  }
  if (no_roots != deg_sigma) return -1;                       /* :405-415 */

  /* omega :419-434 */
  int deg_omega = 0;
  for (int i = 0; i < 2 * RS_T; i++) {
    int tmp = 0;
    int j = (deg_sigma < i) ? deg_sigma : i;
    for (; j >= 0; j--) tmp ^= g_mul(rs, syn[i - j], sigma[j]);
    if (tmp) deg_omega = i;
    omega[i] = (unsigned char)tmp;
  }
  omega[2 * RS_T] = 0;
  /* The reference declares omega[2t] and still executes the line above (:255,:434): a
   * one-byte stack overflow that, as compiled by gcc 11 / clang 22 on x86-64, lands on
   * loc[0].  compat=1 reproduces that observed behaviour; compat=0 is the intended decoder. */
  if (compat) loc[0] = 0;

  /* Forney :445-486 */
  for (int j = no_roots - 1; j >= 0; j--) {
    int num1 = 0;
    for (int i = deg_omega; i >= 0; i--) num1 ^= g_pow(rs, omega[i], i * root[j]);
    int num2 = g_exp(rs, RS_N - root[j]);
    int den = 0;
    int deg_max = deg_sigma < 2 * RS_T - 1 ? deg_sigma : 2 * RS_T - 1;
    for (int i = 1; i <= deg_max; i += 2)
      if (sigma[i]) den ^= g_exp(rs, rs->log[sigma[i]] + (i - 1) * root[j]);
    if (den == 0) return -1;
    int err = g_div(rs, g_mul(rs, num1, num2), den);
    data[loc[j]] ^= (unsigned char)err;
  }
  return no_roots;
}

/* lib/reed_solomon_dec_impl.cc:77-116 */
void o_rs_dec_block(const o_rs *rs, const unsigned char *in, unsigned char *out,
                    size_t nwords, int compat, int *nfail, int *ncorr)
{
  unsigned char w[RS_N];
  int f = 0, c = 0;
  for (size_t i = 0; i < nwords; i++) {
    memset(w, 0, RS_S);
    memcpy(w + RS_S, in + i * (RS_N - RS_S), RS_N - RS_S);
    int r = o_rs_decode(rs, w, compat);
    if (r < 0) f++; else c += r;
    memcpy(out + i * (RS_K - RS_S), w + RS_S, RS_K - RS_S);   /* output regardless of r */
  }
  if (nfail) *nfail = f;
  if (ncorr) *ncorr = c;
}
