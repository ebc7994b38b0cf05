// CPU check of the TPS word's BCH test as the segment-parallel bookkeeping runs it (gr_dvbt_amd/csrc/k_frontend.hpp: bch_check_tab on the table of
// tps_bch_table_host, extracted by tests/test_tps_bch_host.py) against the oracle's bit-serial restatement of verify_bch_code
// (oracle/o_config.c:o_bch_check, lib/reference_signals_impl.cc:385-425): formatted TPS words of every frame and configuration, the same with 1..4 flipped
// bits, and random 68-bit words.
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#define __device__
#define __forceinline__ inline
#include "bch_fn.inc"
extern "C" {   // oracle/dvbt_oracle.h is a C header (C99 complex types); the three functions used here, with o_cfg as an opaque block
void o_cfg_init(void *c, int constellation, int hierarchy, int code_rate, int guard, int mode, int include_cell_id, int cell_id);
void o_tps_format(const void *c, int frame_index, const char *wk, unsigned char *tps68);
int o_bch_check(const unsigned char *tps68);
}
int main()
{
  const std::vector<uint16_t> T = tps_bch_table_host();
  srand(5);
  long total = 0, bad = 0, valid = 0;
  auto check = [&](const unsigned char *t) {
    unsigned long long lo = 0; unsigned hi = 0;
    for (int i = 0; i < 64; i++) lo |= (unsigned long long)(t[i] & 1) << i;
    for (int i = 64; i < 68; i++) hi |= (unsigned)(t[i] & 1) << (i - 64);
    const int a = bch_check_tab(T.data(), lo, hi), b = o_bch_check(t);
    total++; if (a != b) bad++; if (b == 0) valid++;
  };
  for (int constellation = 0; constellation < 3; constellation++)
    for (int rate = 0; rate < 5; rate++)
      for (int mode = 0; mode < 2; mode++)
        for (int guard = 0; guard < 4; guard++)
          for (int hier = 0; hier < 4; hier++)
            for (int frame = 0; frame < 4; frame++) {
              alignas(16) static unsigned char c[4096]; o_cfg_init(c, constellation, hier, rate, guard, mode, frame & 1, (0x12 * frame) & 0xff);
              unsigned char t[68];
              const char wk[1] = {(char)(rand() & 1)};
              o_tps_format(c, frame, wk, t);
              check(t);
              for (int flips = 1; flips <= 4; flips++) {
                unsigned char u[68]; memcpy(u, t, 68);
                for (int k = 0; k < flips; k++) u[rand() % 68] ^= 1;
                check(u);
              }
            }
  for (int i = 0; i < 200000; i++) { unsigned char t[68]; for (int k = 0; k < 68; k++) t[k] = rand() & 1; check(t); }
  printf("%ld words, %ld valid, %ld mismatches\n", total, valid, bad);
  return bad != 0;
}
