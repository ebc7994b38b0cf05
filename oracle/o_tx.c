/* o_tx.c -- loopback baseband generator (the TX chain as far as test vectors need it).
 * TEST INFRASTRUCTURE / test utility, not a performance target.
 * Restates: energy_dispersal (lib/energy_dispersal_impl.cc:106-141), reed_solomon_enc
 * (lib/reed_solomon_enc_impl.cc:84-92), convolutional_interleaver
 * (lib/convolutional_interleaver_impl.cc:73-82), inner_coder (lib/inner_coder_impl.cc:33-121,
 * 225-254), bit_inner_interleaver, symbol_inner_interleaver(dir=1), dvbt_map
 * (lib/dvbt_map_impl.cc:162-163), reference_signals / pilot_gen::update_output
 * (lib/reference_signals_impl.cc:1127-1186, get_tpilot_value :831-842), then
 * fft_vxx(reverse, shift) + cyclic prefixer + multiply_const as in apps/dvbt_tx_demo*.grc. */
#include "dvbt_oracle.h"
#include <stdlib.h>
#include <string.h>

/* info bits carried by one OFDM symbol (an integer number of puncture periods, not
 * necessarily a whole number of bytes: the reference's inner_coder works on groups of
 * 4 symbols for that reason, set_output_multiple(4), inner_coder_impl.cc:160-172) */
static size_t info_bits_per_symbol(const o_cfg *c)
{ return (size_t)c->payload * c->m * c->k / c->n; }

size_t o_tx_symbols_for_packets(const o_cfg *c, size_t npackets)
{ return (npackets * 204 * 8) / info_bits_per_symbol(c); }

/* inner_coder_impl.cc:33-48: one mother-code step; reg persists */
static void codeword(unsigned *reg, int bit, int *x, int *y)
{
  *reg |= ((unsigned)bit & 1) << 7;
  *reg >>= 1;
  unsigned r = *reg;
  *x = ((r >> 6) ^ (r >> 5) ^ (r >> 4) ^ (r >> 3) ^ r) & 1;   /* G1 = 171 oct */
  *y = ((r >> 6) ^ (r >> 4) ^ (r >> 3) ^ (r >> 1) ^ r) & 1;   /* G2 = 133 oct */
}

size_t o_tx_generate(const o_cfg *c, const unsigned char *ts, size_t npackets, float scale,
                     ocf *iq, size_t cap, ocf *freq_taps)
{ return o_tx_generate_from(c, ts, npackets, 0, scale, iq, cap, freq_taps); }

/* packet0: index of ts[0] in the whole TS when the generator starts at a later superframe of a long stream (a multiple
 * of the packets per superframe): only the phase of the 8-packet energy-dispersal groups depends on it.  The byte
 * interleaver and the convolutional encoder start from zero state, so the first 11 RS words' worth of symbols differ
 * from the same symbols of a generator that started at packet 0; everything after is identical. */
size_t o_tx_generate_from(const o_cfg *c, const unsigned char *ts, size_t npackets, size_t packet0, float scale,
                          ocf *iq, size_t cap, ocf *freq_taps)
{
  const int N = c->N, cp = c->cp, K = c->Kmax + 1, zl = c->zeros_left;
  const size_t ibits = info_bits_per_symbol(c);
  size_t bitpos = 0;
  size_t nsym = o_tx_symbols_for_packets(c, npackets);
  if (nsym * (size_t)(N + cp) > cap) nsym = cap / (size_t)(N + cp);

  /* outer coder */
  o_rs rs; o_rs_init(&rs);
  size_t nbytes = npackets * 204;
  unsigned char *disp = malloc(npackets * 188), *rsout = malloc(nbytes), *il = malloc(nbytes);
  o_energy_dispersal_from(ts, disp, npackets, packet0);
  unsigned char w[255];
  for (size_t p = 0; p < npackets; p++) {
    memset(w, 0, 51); memcpy(w + 51, disp + p * 188, 188);
    o_rs_encode(&rs, w, w + 239);
    memcpy(rsout + p * 204, w + 51, 204);
  }
  o_conv_interleave(rsout, il, nbytes);

  /* inner coder: puncture pattern == the depuncture vector of the RX side */
  int plen; const unsigned char *punct = o_vit_puncture(c->code_rate, &plen);
  size_t coded_per_sym = (size_t)c->payload * c->m;
  unsigned char *cbits = malloc(coded_per_sym + 16);
  unsigned char *sym_a = malloc(c->payload), *sym_b = malloc(c->payload);
  int *H = malloc(sizeof(int) * c->payload); o_sym_H(c, H);
  ocf *points = malloc(sizeof(ocf) * c->csize); o_constellation(c, 1.0f, points);
  char *wk = malloc(K); o_prbs_wk(c, wk);
  ocf *freq = malloc(sizeof(ocf) * N), *timeb = malloc(sizeof(ocf) * N);
  ocf *tps_val = calloc(c->n_tps, sizeof(ocf));
  unsigned char tps_data[68];
  unsigned enc_reg = 0; size_t pphase = 0;
  int symbol_index = 0, frame_index = 0;

  for (size_t s = 0; s < nsym; s++) {
    /* encode ib bytes, MSB first (inner_coder_impl.cc:225-254) */
    size_t nb = 0;
    for (size_t i = 0; i < ibits; i++, bitpos++) {
      int x, y; codeword(&enc_reg, (il[bitpos >> 3] >> (7 - (bitpos & 7))) & 1, &x, &y);
      if (punct[pphase]) cbits[nb++] = (unsigned char)x;
      if (punct[pphase + 1]) cbits[nb++] = (unsigned char)y;
      pphase = (pphase + 2) % (size_t)plen;
    }
    for (int i = 0; i < c->payload; i++) {
      int v = 0;
      for (int j = 0; j < c->m; j++) v |= cbits[(size_t)c->m * i + j] << (c->m - 1 - j);
      sym_a[i] = (unsigned char)v;
    }
    o_bit_interleave(c, sym_a, sym_b, c->payload);
    o_sym_interleave(c, H, sym_b, sym_a, symbol_index, 1);

    /* pilot_gen::update_output :1127-1186 */
    o_tps_format(c, frame_index, wk, tps_data);
    memset(freq, 0, sizeof(ocf) * N);
    int sp = 0, cpi = 0, tpi = 0, pc = 0;
    int sp_size = c->n_spilot + (symbol_index == 0 ? 1 : 0);      /* advance_spilot uses sindex==0 */
    for (int k = 0; k < K; k++) {
      int is_payload = 1;
      if (k == 3 * (symbol_index % 4) + 12 * sp) {
        freq[zl + k] = (float)(4 * 2 * (0.5 - wk[k]) / 3); sp = (sp + 1) % sp_size; is_payload = 0;
      }
      if (k == c->cpilot[cpi]) {
        freq[zl + k] = (float)(4 * 2 * (0.5 - wk[k])) / 3; cpi = (cpi + 1) % c->n_cpilot; is_payload = 0;
      }
      if (k == c->tps[tpi]) {
        if (symbol_index == 0) tps_val[tpi] = (float)(2 * (0.5 - wk[k]));
        else if (tps_data[symbol_index] == 1) tps_val[tpi] = -crealf(tps_val[tpi]);
        freq[zl + k] = tps_val[tpi]; tpi = (tpi + 1) % c->n_tps; is_payload = 0;
      }
      if (is_payload) freq[zl + k] = points[sym_a[pc++]];
    }
    if (++symbol_index == 68) { symbol_index = 0; if (++frame_index == 4) frame_index = 0; }

    if (freq_taps) memcpy(freq_taps + s * N, freq, sizeof(ocf) * N);
    o_ifft_shift(N, freq, timeb);
    ocf *o = iq + s * (size_t)(N + cp);
    for (int i = 0; i < cp; i++) o[i] = scale * timeb[N - cp + i];
    for (int i = 0; i < N; i++) o[cp + i] = scale * timeb[i];
  }
  free(disp); free(rsout); free(il); free(cbits); free(sym_a); free(sym_b); free(H);
  free(points); free(wk); free(freq); free(timeb); free(tps_val);
  return nsym * (size_t)(N + cp);
}
