/* o_refbench.c -- times the REFERENCE's own SSE2 Viterbi kernels (oracle/_ref/libdviterbi_ref.so = lib/d_viterbi.c + d_tab.c +
 * d_metrics.c compiled unmodified, oracle/Makefile) in the block's calling pattern (viterbi_decoder_impl.cc:261-292), natively.
 * TEST INFRASTRUCTURE (see dvbt_oracle.h): bench.py's cpu_baseline leg reports it beside the port's rate. */
#include "dvbt_oracle.h"
#include <dlfcn.h>
#include <stdlib.h>
#include <time.h>

typedef void (*init_fn)(void *mm0, void *pp0);
typedef void (*bfly_fn)(unsigned char *symbols, void *m0, void *m1, void *p0, void *p1);
typedef unsigned char (*out_fn)(void *mm0, void *pp0, int ntraceback, unsigned char *outbuf);

/* decoded Mbit/s over nsym depunctured symbols (2 per trellis step), -1 when the library is missing */
double o_ref_viterbi_mbps(const char *so_path, size_t nsym, int ntraceback)
{
  void *h = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return -1.0;
  init_fn init = (init_fn)dlsym(h, "d_viterbi_chunks_init_sse2");
  bfly_fn bfly = (bfly_fn)dlsym(h, "d_viterbi_butterfly2_sse2");
  out_fn outp = (out_fn)dlsym(h, "d_viterbi_get_output_sse2");
  if (!init || !bfly || !outp) { dlclose(h); return -1.0; }
  unsigned char *sym = malloc(nsym + 16);
  unsigned x = 777;
  for (size_t i = 0; i < nsym; i++) { x = x * 1664525u + 1013904223u; sym[i] = (unsigned char)((x >> 24) & 1); }
  unsigned char *st = NULL;
  if (posix_memalign((void **)&st, 16, 4 * 64)) { free(sym); dlclose(h); return -1.0; }
  unsigned char *m0 = st, *m1 = st + 64, *p0 = st + 128, *p1 = st + 192, ch = 0, acc = 0;
  init(m0, p0);
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (size_t ic = 0; ic + 4 <= nsym; ic += 4) {
    bfly(sym + ic, m0, m1, p0, p1);
    if (ic > 0 && (ic % 16) == 8) { outp(m0, p0, ntraceback, &ch); acc ^= ch; }
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  sym[0] = acc;
  free(sym); free(st); dlclose(h);
  return (double)(nsym / 2) / dt / 1e6;
}

/* viterbi_decoder_impl::general_work (lib/viterbi_decoder_impl.cc:241-292) around the REFERENCE's own kernels, natively, for a stream that starts with a reset, over nsym input
 * bytes (a whole number of blocks): the depuncturer and the butterfly / output cadence exactly as o_viterbi_decode_n (o_viterbi.c) drives the restatement, so that the two can be
 * compared on megabytes (tests/test_oracle_ref_viterbi.py; the Python loop over the kernels manages kilobytes).  Returns the bytes written, or (size_t)-1 when the library is missing. */
size_t o_ref_viterbi_decode_n(const char *so_path, const o_cfg *c, const unsigned char *in, size_t nsym, unsigned char *out)
{
  void *h = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return (size_t)-1;
  init_fn init = (init_fn)dlsym(h, "d_viterbi_chunks_init_sse2");
  bfly_fn bfly = (bfly_fn)dlsym(h, "d_viterbi_butterfly2_sse2");
  out_fn outp = (out_fn)dlsym(h, "d_viterbi_get_output_sse2");
  unsigned char *st = NULL;
  if (!init || !bfly || !outp || posix_memalign((void **)&st, 16, 4 * 64 + 16)) { dlclose(h); return (size_t)-1; }
  unsigned char *m0 = st, *m1 = st + 64, *p0 = st + 128, *p1 = st + 192, *bits = st + 256;
  int plen; const unsigned char *punct = o_vit_puncture(c->code_rate, &plen);
  const int nt = o_vit_ntraceback(c->code_rate);
  init(m0, p0);
  size_t out_count = 0, count = 0, ic = 0; int nb = 0;
#define R_PUSH(b) do { bits[nb++] = (unsigned char)(b); count++; if (nb == 4) { nb = 0; \
    bfly(bits, m0, m1, p0, p1); \
    if (ic > 0 && (ic % 16) == 8) { unsigned char ch = 0; outp(m0, p0, nt, &ch); if (out_count >= (size_t)nt) out[out_count - nt] = ch; out_count++; } \
    ic += 4; } } while (0)
  for (size_t i = 0; i < nsym; i++)
    for (int j = c->m - 1; j >= 0; j--) {
      while (punct[count % (size_t)(2 * c->k)] == 0) R_PUSH(2);
      R_PUSH((in[i] >> j) & 1);
      while (punct[count % (size_t)(2 * c->k)] == 0) R_PUSH(2);
    }
#undef R_PUSH
  free(st); dlclose(h);
  return out_count >= (size_t)nt ? out_count - nt : 0;
}
