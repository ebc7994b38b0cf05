/* o_outer.c -- Forney byte (de)interleaver and energy dispersal/descramble.
 * TEST INFRASTRUCTURE (see dvbt_oracle.h). */
#include "dvbt_oracle.h"
#include <string.h>

/* lib/convolutional_deinterleaver_impl.cc:64-65,133-138: branch j=n%12 is a FIFO of
 * 17*(11-j) zeros -> out[n] = in[n - 204*(11 - n%12)], zero before the start. */
void o_conv_deinterleave(const unsigned char *in, unsigned char *out, size_t n)
{
  for (size_t i = 0; i < n; i++) {
    size_t d = 204u * (11u - (unsigned)(i % 12));
    out[i] = i >= d ? in[i - d] : 0;
  }
}

/* lib/convolutional_interleaver_impl.cc:49-50,73-82: branch j delay 17*j */
void o_conv_interleave(const unsigned char *in, unsigned char *out, size_t n)
{
  for (size_t i = 0; i < n; i++) {
    size_t d = 204u * (unsigned)(i % 12);
    out[i] = i >= d ? in[i - d] : 0;
  }
}

/* PRBS 1+x^14+x^15, init 0xa9: lib/energy_dispersal_impl.cc:37-58 (== energy_descramble_impl.cc:46-67) */
static int clock_prbs(unsigned *reg, int clocks)
{
  int res = 0;
  for (int i = 0; i < clocks; i++) {
    int fb = ((*reg >> 13) ^ (*reg >> 14)) & 1;
    *reg = ((*reg << 1) | (unsigned)fb) & 0x7fff;
    res = (res << 1) | fb;
  }
  return res;
}

/* xor mask for one group of 8 packets (1504 bytes); sync positions hold 0 */
void o_energy_prbs(unsigned char *seq)
{
  unsigned reg = 0xa9;
  int count = 0;
  for (int p = 0; p < 8; p++) {
    seq[count++] = 0;
    for (int k = 1; k < 188; k++) seq[count++] = (unsigned char)clock_prbs(&reg, 8);
    clock_prbs(&reg, 8);
  }
}

/* lib/energy_dispersal_impl.cc:106-141 for a TS that starts on a sync byte; packet0 = index in the whole TS of
 * ts[0] (the 8-packet groups count from the start of the whole TS) */
void o_energy_dispersal_from(const unsigned char *ts, unsigned char *out, size_t npackets, size_t packet0)
{
  unsigned char seq[1504];
  o_energy_prbs(seq);
  for (size_t p = 0; p < npackets; p++) {
    size_t g = (p + packet0) % 8;
    out[p * 188] = g == 0 ? 0xB8 : 0x47;
    for (int k = 1; k < 188; k++) out[p * 188 + k] = ts[p * 188 + k] ^ seq[g * 188 + k];
  }
}

void o_energy_dispersal(const unsigned char *ts, unsigned char *out, size_t npackets)
{ o_energy_dispersal_from(ts, out, npackets, 0); }

/* the descrambling of :143-163 alone: ngroups groups of 8 packets that start on an NSYNC packet */
size_t o_energy_descramble_groups(const unsigned char *in, size_t ngroups, unsigned char *out)
{
  unsigned char seq[1504];
  o_energy_prbs(seq);
  size_t written = 0;
  for (size_t i = 0; i < ngroups; i++)
    for (int k = 0; k < 1504; k++) {
      unsigned char b = in[i * 1504 + k];
      out[written++] = (k % 188 == 0) ? 0x47 : (b ^ seq[k]);
    }
  return written;
}

/* lib/energy_descramble_impl.cc:108-174 over the whole RS output, call by call in the smallest calls the block accepts
 * (noutput_items = 4 items' worth: 4 items visible, 2 consumed, :90,:139-141): every call first checks the byte at its
 * offset d_index and searches on at 188-byte strides within the first two items when it is not an NSYNC (:121-123); a
 * search that fails drops two items and starts over at offset 0 (:129-134).  On a stream without sync errors this is:
 * lock on the first NSYNC, descramble everything from there, hold back the last two items. */
size_t o_energy_descramble(const unsigned char *in, size_t nitems, unsigned char *out)
{
  unsigned char seq[1504];
  o_energy_prbs(seq);
  const size_t d_search = 2 * 1504;
  size_t base = 0, d_index = 0, written = 0;      /* base: items consumed so far */
  while (nitems - base >= 4) {
    const unsigned char *p = in + base * 1504;
    while (d_index < d_search && p[d_index] != 0xB8) d_index += 188;
    if (d_index >= d_search) { d_index = 0; base += 2; continue; }
    for (size_t i = 0; i < 2; i++)
      for (int k = 0; k < 1504; k++) {
        unsigned char b = p[d_index + i * 1504 + k];
        out[written++] = (k % 188 == 0) ? 0x47 : (b ^ seq[k]);
      }
    base += 2;
  }
  return written;
}
