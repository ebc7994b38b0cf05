/* o_inner.c -- constellation, demapper, symbol and bit (de)interleavers.
 * TEST INFRASTRUCTURE (see dvbt_oracle.h). */
#include "dvbt_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

/* ---------------- constellation: lib/dvbt_demap_impl.cc:117-165 (== dvbt_map_impl.cc:91-139) */
static int bin_to_gray(int v) { return (v >> 1) ^ v; }

void o_constellation(const o_cfg *c, float gain_in, ocf *points)
{
  float gain = gain_in * c->norm;                      /* dvbt_demap_impl.cc:72 */
  int size = c->csize;
  int bits_per_axis = c->m / 2;
  int steps_per_axis = (int)(sqrt((double)size) / 2 - 1);
  for (int i = 0; i < size; i++) {
    int q = (i >> (2 * (bits_per_axis - 1))) & 3;
    int sign0 = (q >> 1) ? -1 : 1, sign1 = (q & 1) ? -1 : 1;
    int x = (i >> (bits_per_axis - 1)) & ((1 << (bits_per_axis - 1)) - 1);
    int y = i & ((1 << (bits_per_axis - 1)) - 1);
    int xval = c->alpha + (steps_per_axis - x) * c->step;
    int yval = c->alpha + (steps_per_axis - y) * c->step;
    int val = (bin_to_gray(x) << (bits_per_axis - 1)) + bin_to_gray(y);
    x = 0; y = 0;
    for (int j = 0; j < bits_per_axis - 1; j++) {
      x += ((val >> (1 + 2 * j)) & 1) << j;
      y += ((val >> (2 * j)) & 1) << j;
    }
    val = (q << (2 * (bits_per_axis - 1))) + (x << (bits_per_axis - 1)) + y;
    /* d_gain * gr_complex(int,int): float * complex<float> */
    points[val] = gain * (float)(sign0 * xval) + I * (gain * (float)(sign1 * yval));
  }
}

/* lib/dvbt_demap_impl.cc:167-203: first strict minimum of |v-p|^2 over the table.
 * VOLK generic square_dist: (re diff)^2 + (im diff)^2 in float. */
void o_demap(const o_cfg *c, const ocf *points, const ocf *in, unsigned char *out, size_t n)
{
  for (size_t i = 0; i < n; i++) {
    float vr = crealf(in[i]), vi = cimagf(in[i]);
    float dr = vr - crealf(points[0]), di = vi - cimagf(points[0]);
    float min_dist = dr * dr + di * di;
    int min_index = 0;
    for (int j = 0; j < c->csize; j++) {
      dr = vr - crealf(points[j]); di = vi - cimagf(points[j]);
      float d = dr * dr + di * di;
      if (d < min_dist) { min_dist = d; min_index = j; }
    }
    out[i] = (unsigned char)min_index;
  }
}

/* ---------------- symbol interleaver: lib/symbol_inner_interleaver_impl.cc:32-96,161-219 */
static const char bit_perm_2k[] = { 4, 3, 9, 6, 2, 8, 1, 5, 7, 0 };
static const char bit_perm_8k[] = { 7, 1, 4, 2, 9, 6, 8, 10, 0, 3, 11, 5 };

void o_sym_H(const o_cfg *c, int *h)
{
  const int Nr = c->mode == O_T8k ? 13 : 11;
  const char *perm = c->mode == O_T8k ? bit_perm_8k : bit_perm_2k;
  const int mask = (1 << Nr) - 1;
  int q = 0, reg = 0;
  /* calculate_R(i) re-runs the LFSR from scratch for every i (:56-96); the register after
   * i-2 clocks is the same value, so it is advanced incrementally here. */
  for (int i = 0; i < c->N && q < c->payload; i++) {
    if (i < 2) reg = 0;
    else if (i == 2) reg = 1;
    else {
      int nb = c->mode == O_T8k ? (reg ^ (reg >> 1) ^ (reg >> 4) ^ (reg >> 6)) & 1
                                : (reg ^ (reg >> 3)) & 1;
      reg = ((reg >> 1) | (nb << (Nr - 2))) & mask;
    }
    int newreg = 0;
    for (int k = 0; k < Nr - 1; k++) newreg |= ((reg >> k) & 1) << perm[k];
    int v = ((i % 2) << (Nr - 1)) + newreg;
    if (v < c->payload) h[q++] = v;
  }
}

void o_sym_interleave(const o_cfg *c, const int *h, const unsigned char *in,
                      unsigned char *out, int symbol_index, int direction)
{
  int odd = symbol_index % 2;
  if (direction) {           /* TX :182-195 */
    for (int q = 0; q < c->payload; q++)
      if (odd) out[q] = in[h[q]]; else out[h[q]] = in[q];
  } else {                   /* RX :197-209 */
    for (int q = 0; q < c->payload; q++)
      if (odd) out[h[q]] = in[q]; else out[q] = in[h[q]];
  }
}

/* ---------------- bit (de)interleaver, non-hierarchical
 * lib/bit_inner_interleaver_impl.cc:137-176, lib/bit_inner_deinterleaver_impl.cc:120-184 */
static int Hbit(int e, int w)
{
  static const int off[6] = { 0, 63, 105, 42, 21, 84 };
  return (w + off[e]) % 126;
}
static int perm_nh(int v, int i) { return ((i % v) / (v / 2)) + 2 * (i % (v / 2)); }

void o_bit_interleave(const o_cfg *c, const unsigned char *in, unsigned char *out, size_t n)
{
  int v = c->m;
  unsigned char b[6][126];
  for (size_t blk = 0; blk < n / 126; blk++) {
    for (int i = 0; i < 126; i++) {
      int ch = in[blk * 126 + i];
      for (int k = 0; k < v; k++) b[perm_nh(v, v * i + k)][i] = (ch >> (v - k - 1)) & 1;
    }
    for (int w = 0; w < 126; w++) {
      int val = 0;
      for (int e = 0; e < v; e++) val = (val << 1) | b[e][Hbit(e, w)];
      out[blk * 126 + w] = (unsigned char)val;
    }
  }
}

void o_bit_deinterleave(const o_cfg *c, const unsigned char *in, unsigned char *out, size_t n)
{
  int v = c->m;
  unsigned char b[6][126];
  for (size_t blk = 0; blk < n / 126; blk++) {
    for (int w = 0; w < 126; w++) {
      int ch = in[blk * 126 + w];
      for (int e = 0; e < v; e++) b[e][Hbit(e, w)] = (ch >> (v - e - 1)) & 1;
    }
    for (int i = 0; i < 126; i++) {
      int val = 0;
      for (int k = 0; k < v; k++) val = (val << 1) | b[perm_nh(v, v * i + k)][i];
      out[blk * 126 + i] = (unsigned char)val;
    }
  }
}

/* ---------------- bit de-interleaver, hierarchical: lib/bit_inner_deinterleaver_impl.cc:91-99,148-184, two output streams.
 * The reference addresses its bit matrix d_b[v][126] with second indices far beyond 125 (:169 `(d_v * i + k) / 2`, :176 `(d_v * i + k) / (d_v - 2)`): C lays
 * the matrix out row after row, so d_b[e][j] IS flat element e * 126 + j -- inside the matrix for every HP bit and for the LP bits of rows 2..4.
 * Row 5 (64-QAM) with j >= 126 (i >= 84) lies BEHIND the matrix: the reference reads whatever the stack holds there (undefined behaviour, not
 * reproducible); this restatement -- and the product -- take 0 for those bits and say so.  The LP loop runs k = 2 .. d_v - 3 (:175), i.e. not at all for
 * 16-QAM (outl = 0) and over two bits for 64-QAM, with d_perm of :91-99's hierarchical branch. */
static int perm_h(int v, int i) { return (i % (v - 2)) / ((v - 2) / 2) + 2 * (i % ((v - 2) / 2)) + 2; }

void o_bit_deinterleave_hier(const o_cfg *c, const unsigned char *in, unsigned char *outh, unsigned char *outl, size_t n)
{
  int v = c->m;
  unsigned char b[6 * 126];
  for (size_t blk = 0; blk < n / 126; blk++) {
    for (int w = 0; w < 126; w++) {
      int ch = in[blk * 126 + w];
      for (int e = 0; e < v; e++) b[e * 126 + Hbit(e, w)] = (ch >> (v - e - 1)) & 1;
    }
    for (int i = 0; i < 126; i++) {
      int val = 0;
      for (int k = 0; k < 2; k++) val = (val << 1) | b[((v * i + k) % 2) * 126 + (v * i + k) / 2];
      outh[blk * 126 + i] = (unsigned char)val;
      val = 0;
      for (int k = 2; k < v - 2; k++) {
        int f = perm_h(v, v * i + k) * 126 + (v * i + k) / (v - 2);
        val = (val << 1) | (f < v * 126 ? b[f] : 0);              /* behind the matrix: undefined in the reference, 0 here */
      }
      outl[blk * 126 + i] = (unsigned char)val;
    }
  }
}
