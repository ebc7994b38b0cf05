// CPU check of rs_decode_word_lane (gr_dvbt_amd/csrc/k_backend.hpp: the RS(255,239) decoder with one word per lane) against the oracle's
// rs_decode restatement: the function's source is extracted from the header by tests/test_rs_lane_host.py (lane_fn.inc) and compiled for one lane
// (the wavefront votes and shuffles become identities: they only widen loop bounds).  0..13 symbol errors, heavy garbage, both decoder modes.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#define __device__
constexpr int RS_SYN_STRIDE = 65;
#define __any(x) (x)
#define __all(x) (x)
#define __shfl_xor(v, o) (v)
#include "lane_fn.inc"
extern "C" {
typedef struct { unsigned char exp[256], log[256], l[256], g[17]; } o_rs;
void o_rs_init(o_rs *rs);
void o_rs_encode(const o_rs *rs, const unsigned char *data239, unsigned char *parity16);
int o_rs_decode(const o_rs *rs, unsigned char *data255, int compat);
}
int main()
{
  o_rs rs; o_rs_init(&rs);
  uint8_t ex[512], lg[256];
  { int reg = 1; lg[0] = 255; for (int i = 0; i < 255; i++) { ex[i] = reg; ex[i + 255] = reg; lg[reg] = i; reg <<= 1; if (reg & 0x100) reg ^= 0x11d; reg &= 0xff; } ex[510] = ex[0]; ex[511] = ex[1]; }
  srand(7);
  int bad = 0, total = 0;
  for (int trial = 0; trial < 20000; trial++) {
    uint8_t w[255]; memset(w, 0, 51);
    for (int i = 51; i < 239; i++) w[i] = rand() & 255;
    o_rs_encode(&rs, w, w + 239);
    int ne = trial % 14;                       // 0..13 errors
    if (trial % 97 == 0) ne = 1 + rand() % 40; // heavy garbage sometimes
    for (int e = 0; e < ne; e++) w[51 + rand() % 204] ^= 1 + rand() % 255;
    for (int compat = 0; compat < 2; compat++) {
      uint8_t a[255], b[255]; memcpy(a, w, 255); memcpy(b, w, 255);
      int r0 = o_rs_decode(&rs, a, compat);
      // syndromes as the kernel computes them: S_i = C(alpha^i)
      uint8_t syn[16 * 65]; memset(syn, 0, sizeof syn);
      int any = 0;
      for (int i = 0; i < 16; i++) { int sv = 0; for (int j = 0; j < 255; j++) { sv = (sv ? ex[(lg[sv] + i) % 255] : 0) ^ b[j]; } syn[i * 65] = sv; any |= sv; }
      uint8_t root[17 * 64];
      int r1 = 0;
      if (any) r1 = rs_decode_word_lane(b + 51, syn, ex, lg, compat, root, true);
      total++;
      if (r0 != r1 || memcmp(a + 51, b + 51, 188) != 0) { if (bad < 10) printf("MISMATCH trial %d ne %d compat %d: ref %d lane %d\n", trial, ne, compat, r0, r1); bad++; }
    }
  }
  printf("%d mismatches of %d\n", bad, total);
  return bad != 0;
}
