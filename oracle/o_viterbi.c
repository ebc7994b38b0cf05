/* o_viterbi.c -- depuncture + K=7 hard-decision Viterbi with 8-step path bytes.
 * TEST INFRASTRUCTURE (see dvbt_oracle.h).
 * Restates lib/viterbi_decoder_impl.cc:61-65,95-124,149-153,192-324 and the SSE2 kernels
 * lib/d_viterbi.c:261-285 (init), :461-576 (butterfly2), :680-735 (get_output),
 * parity table lib/d_tab.c:24-57.  The 16-byte SSE2 lanes are written as plain 64-entry
 * uint8 arrays in the same state order (state s = byte s), so gcc -O3 -msse2 vectorises it.
 * Pinned bit-for-bit against oracle/_ref (the reference's own d_viterbi.c) by
 * tests/test_oracle_ref_viterbi.py. */
#include "dvbt_oracle.h"
#include <string.h>
#include <stdlib.h>

static unsigned char branchtab[2][32];
static int branch_ready = 0;

static int parity8(int x) { x ^= x >> 4; x ^= x >> 2; x ^= x >> 1; return x & 1; }

static void branch_init(void)
{ /* d_viterbi.c:272-277: POLYA=0x4f, POLYB=0x6d */
  if (branch_ready) return;
  for (int i = 0; i < 32; i++) {
    branchtab[0][i] = (unsigned char)parity8((2 * i) & 0x4f);
    branchtab[1][i] = (unsigned char)parity8((2 * i) & 0x6d);
  }
  branch_ready = 1;
}

void o_vit_core_init(o_vit_core *v, int ntraceback)
{
  branch_init();
  int sp = v->store_pos;           /* store_pos is NOT reset by the reference (B-2) */
  memset(v, 0, sizeof *v);
  v->store_pos = sp;
  v->ntraceback = ntraceback;
}

/* one trellis step; d_viterbi.c:483-524 (and its copy :533-575).  Written as the SSE2 code is
 * organised (branch metrics for the 32 butterflies first, then the ACS, then the interleave of
 * even/odd successors) with every loop over plain uint8 arrays so that gcc -O3 -msse2 vectorises it. */
static void step(const unsigned char *metric0, const unsigned char *path0,
                 unsigned char *metric1, unsigned char *path1,
                 unsigned char s0, unsigned char s1)
{
  unsigned char metsvm[32], metsv[32], sv0[32], sv1[32], pt0[32], pt1[32];
  if (s0 == 2)      for (int b = 0; b < 32; b++) { metsvm[b] = branchtab[1][b] ^ s1; metsv[b] = (unsigned char)(1 - metsvm[b]); }
  else if (s1 == 2) for (int b = 0; b < 32; b++) { metsvm[b] = branchtab[0][b] ^ s0; metsv[b] = (unsigned char)(1 - metsvm[b]); }
  else for (int b = 0; b < 32; b++) { metsvm[b] = (unsigned char)((branchtab[0][b] ^ s0) + (branchtab[1][b] ^ s1)); metsv[b] = (unsigned char)(2 - metsvm[b]); }
  for (int b = 0; b < 32; b++) {
    unsigned char m0 = (unsigned char)(metric0[b] + metsv[b]);
    unsigned char m1 = (unsigned char)(metric0[b + 32] + metsvm[b]);
    unsigned char m2 = (unsigned char)(metric0[b] + metsvm[b]);
    unsigned char m3 = (unsigned char)(metric0[b + 32] + metsv[b]);
    unsigned char d0 = (signed char)(unsigned char)(m0 - m1) > 0 ? 0xff : 0;
    unsigned char d1 = (signed char)(unsigned char)(m2 - m3) > 0 ? 0xff : 0;
    unsigned char sh0 = (unsigned char)(path0[b] << 1);
    unsigned char sh1 = (unsigned char)((path0[b + 32] << 1) + 1);
    sv0[b] = (unsigned char)((d0 & m0) | (~d0 & m1));
    sv1[b] = (unsigned char)((d1 & m2) | (~d1 & m3));
    pt0[b] = (unsigned char)((d0 & sh0) | (~d0 & sh1));
    pt1[b] = (unsigned char)((d1 & sh0) | (~d1 & sh1));
  }
  for (int b = 0; b < 32; b++) { metric1[2 * b] = sv0[b]; metric1[2 * b + 1] = sv1[b]; path1[2 * b] = pt0[b]; path1[2 * b + 1] = pt1[b]; }
}

/* d_viterbi_butterfly2_sse2: two trellis steps on 4 depunctured symbols */
void o_vit_butterfly2(o_vit_core *v, const unsigned char *sym)
{
  unsigned char m1[64], p1[64];
  step(v->metric, v->path, m1, p1, sym[0], sym[1]);
  step(m1, p1, v->metric, v->path, sym[2], sym[3]);
}

/* d_viterbi_get_output_sse2 :680-735 */
unsigned char o_vit_get_output(o_vit_core *v)
{
  int nt = v->ntraceback;
  v->store_pos = (v->store_pos + 1) % nt;
  memcpy(v->pp[v->store_pos], v->path, 64);
  int best = 0, bestm = v->metric[0], minm = v->metric[0];
  for (int i = 1; i < 64; i++) {
    if (v->metric[i] > bestm) { bestm = v->metric[i]; best = i; }
    if (v->metric[i] < minm) minm = v->metric[i];
  }
  int pos = v->store_pos;
  for (int i = 0; i < nt - 1; i++) {
    best = v->pp[pos][best] >> 2;
    pos = (pos - 1 + nt) % nt;
  }
  unsigned char out = v->pp[pos][best];
  for (int i = 0; i < 64; i++) { v->path[i] = 0; v->metric[i] = (unsigned char)(v->metric[i] - minm); }
  return out;
}

/* viterbi_decoder_impl.cc:61-65 */
static const unsigned char P12[2]  = { 1, 1 };
static const unsigned char P23[4]  = { 1, 1, 0, 1 };
static const unsigned char P34[6]  = { 1, 1, 0, 1, 1, 0 };
static const unsigned char P56[10] = { 1, 1, 0, 1, 1, 0, 0, 1, 1, 0 };
static const unsigned char P78[14] = { 1, 1, 0, 1, 0, 1, 0, 1, 1, 0, 0, 1, 1, 0 };

const unsigned char *o_vit_puncture(int cr, int *len)
{
  switch (cr) {
    case O_C2_3: *len = 4;  return P23;
    case O_C3_4: *len = 6;  return P34;
    case O_C5_6: *len = 10; return P56;
    case O_C7_8: *len = 14; return P78;
    default:     *len = 2;  return P12;
  }
}

int o_vit_ntraceback(int cr)
{ /* viterbi_decoder_impl.cc:95-124 */
  switch (cr) { case O_C2_3: return 9; case O_C3_4: return 10; case O_C5_6: return 15;
                case O_C7_8: return 24; default: return 5; }
}

/* viterbi_decoder_impl.cc:192-324 for a stream that starts with a reset (d_init = 0), over exactly nsym input bytes:
 * the depuncturer and the butterfly/output cadence (:241-292) depend only on the running bit count modulo the puncture
 * period and modulo 16, and a block is a whole number of both, so a run of blocks is one long block.  nsym must
 * make the depunctured bit count a multiple of 16 (any whole number of blocks does). */
/* (snap_at / snaps: test instrumentation, see o_viterbi_decode_snap below; NULL in the decode proper) */
static size_t decode_core(const o_cfg *c, const unsigned char *in, size_t nsym, unsigned char *out,
                          const long long *snap_at, int nsnap, unsigned char *snaps)
{
  int plen; const unsigned char *punct = o_vit_puncture(c->code_rate, &plen);
  int nt = o_vit_ntraceback(c->code_rate);
  o_vit_core v; v.store_pos = 0;
  o_vit_core_init(&v, nt);
  size_t out_count = 0, count = 0, ic = 0;
  unsigned char bits[4]; int nb = 0, si = 0;
  /* depuncture :241-256 feeding the decoder :261-292 four bits (two trellis steps) at a time */
#define O_PUSH(b) do { bits[nb++] = (unsigned char)(b); count++; if (nb == 4) { nb = 0; \
    o_vit_butterfly2(&v, bits); \
    if (ic > 0 && (ic % 16) == 8) { unsigned char ch = o_vit_get_output(&v); if (out_count >= (size_t)nt) out[out_count - nt] = ch; out_count++; \
      if (si < nsnap && (long long)out_count == snap_at[si]) { memcpy(snaps + 64 * (size_t)si, v.metric, 64); si++; } } \
    ic += 4; } } while (0)
  for (size_t i = 0; i < nsym; i++)
    for (int j = c->m - 1; j >= 0; j--) {
      while (punct[count % (size_t)(2 * c->k)] == 0) O_PUSH(2);
      O_PUSH((in[i] >> j) & 1);
      while (punct[count % (size_t)(2 * c->k)] == 0) O_PUSH(2);
    }
#undef O_PUSH
  return out_count >= (size_t)nt ? out_count - nt : 0;
}

/* A decoder that takes up the stream in the state `init` (64 path metrics, minimum subtracted: what o_viterbi_decode_snap reports) right behind get_output call `from_call`
 * of a decode that starts with in[0]: the depuncturer and the output cadence run from in[0] as in decode_core, the trellis only from there on.  out[] is indexed like decode_core's;
 * valid from index from_call - 1 on (the tracebacks of earlier bytes reach into windows this decoder has not seen).  Test instrumentation: the CPU model of the repair passes that
 * make the HIP chunk decoders the streaming decoder (tests/test_viterbi_repair_model.py; k_viterbi3.hpp's viterbi_repair_kernel / viterbi_repair_seq_kernel start decoders the same way). */
size_t o_viterbi_decode_from(const o_cfg *c, const unsigned char *in, size_t nsym, unsigned char *out, long long from_call, const unsigned char *init,
                             const long long *snap_at, int nsnap, unsigned char *snaps)
{
  int plen; const unsigned char *punct = o_vit_puncture(c->code_rate, &plen);
  int nt = o_vit_ntraceback(c->code_rate);
  o_vit_core v; v.store_pos = 0;
  o_vit_core_init(&v, nt);
  size_t out_count = 0, count = 0, ic = 0;
  unsigned char bits[4]; int nb = 0, si = 0, live = from_call <= 0;
  if (live && init) memcpy(v.metric, init, 64);
#define O_PUSH(b) do { bits[nb++] = (unsigned char)(b); count++; if (nb == 4) { nb = 0; \
    if (live) o_vit_butterfly2(&v, bits); \
    if (ic > 0 && (ic % 16) == 8) { \
      if (live) { unsigned char ch = o_vit_get_output(&v); if (out_count >= (size_t)nt) out[out_count - nt] = ch; } \
      out_count++; \
      if (!live && (long long)out_count == from_call) { memcpy(v.metric, init, 64); memset(v.path, 0, 64); live = 1; } \
      if (live && si < nsnap && (long long)out_count == snap_at[si]) { memcpy(snaps + 64 * (size_t)si, v.metric, 64); si++; } } \
    ic += 4; } } while (0)
  for (size_t i = 0; i < nsym; i++)
    for (int j = c->m - 1; j >= 0; j--) {
      while (punct[count % (size_t)(2 * c->k)] == 0) O_PUSH(2);
      O_PUSH((in[i] >> j) & 1);
      while (punct[count % (size_t)(2 * c->k)] == 0) O_PUSH(2);
    }
#undef O_PUSH
  return out_count >= (size_t)nt ? out_count - nt : 0;
}

size_t o_viterbi_decode_n(const o_cfg *c, const unsigned char *in, size_t nsym, unsigned char *out)
{ return decode_core(c, in, nsym, out, NULL, 0, NULL); }

/* The same decode, and the decoder's 64 path metrics as they stand right behind its snap_at[i]-th call of get_output (calls counted from 1; snap_at ascending): get_output has just
 * subtracted their minimum (:728-732) and cleared the path bytes, so two decoders over the same input whose vectors are EQUAL at the same call make identical decisions from there on.
 * tests/test_viterbi_boundary_proof_model.py: the criterion a chunk-parallel decoder could prove its equality with the streaming decoder by, per run. */
size_t o_viterbi_decode_snap(const o_cfg *c, const unsigned char *in, size_t nsym, unsigned char *out, const long long *snap_at, int nsnap, unsigned char *snaps)
{ return decode_core(c, in, nsym, out, snap_at, nsnap, snaps); }

/* block-level entry: whole blocks of bsize*n/m input bytes only (:149,:198) */
size_t o_viterbi_decode(const o_cfg *c, int bsize, const unsigned char *in, size_t nsym,
                        unsigned char *out)
{
  int d_nsymbols = bsize * c->n / c->m;          /* :149 */
  return o_viterbi_decode_n(c, in, (nsym / (size_t)d_nsymbols) * (size_t)d_nsymbols, out);
}
