// rx_stream_example.cpp -- apps/dvbt_rx_demo*.grc from the multiply_const on, with the ten receive blocks replaced by the one block over the
// streaming entry of libdvbt_hip (gr::dvbt::rx_hip; here its GNU Radio-free mirror of dvbt_blocks.hpp): a file of complex64 baseband at the
// OFDM elementary rate is read in scheduler-sized calls (64 OFDM symbols each), the TS is written as it comes out.
//   rx_stream_example <mode 2k|8k> <constellation qpsk|qam16|qam64> <rate 1/2|2/3|3/4|5/6|7/8> <baseband.cf32> <out.ts>
#include <cstdio>
#include <cstring>
#include <vector>
#include "dvbt_blocks.hpp"

using namespace gr::dvbt_amd;

int main(int argc, char **argv)
{
  if (argc < 6) { std::printf("usage: %s <2k|8k> <qpsk|qam16|qam64> <1/2|2/3|3/4|5/6|7/8> <baseband.cf32> <out.ts>\n", argv[0]); return 2; }
  try {
    const dvbt_transmission_mode_t mode = !std::strcmp(argv[1], "8k") ? DVBT_T8k : DVBT_T2k;
    const dvbt_constellation_t con = !std::strcmp(argv[2], "qpsk") ? DVBT_QPSK : !std::strcmp(argv[2], "qam16") ? DVBT_QAM16 : DVBT_QAM64;
    const char *rates[] = {"1/2", "2/3", "3/4", "5/6", "7/8"};
    int cr = 0; for (int i = 0; i < 5; i++) if (!std::strcmp(argv[3], rates[i])) cr = i;
    rx_hip::sptr rx = rx_hip::make(con, DVBT_NH, (dvbt_code_rate_t)cr, DVBT_G1_32, mode, 30.0f, 768, 4);
    std::FILE *f = std::fopen(argv[4], "rb"), *o = std::fopen(argv[5], "wb");
    if (!f || !o) { std::perror("open"); return 1; }
    dvbt_dims d; check(dvbt_get_dims(con, DVBT_NH, cr, DVBT_G1_32, mode, &d));
    const size_t call = (size_t)64 * (d.fft_length + d.cp_length);               // items per work() call
    std::vector<float> in(2 * call); std::vector<unsigned char> out(1 << 22);
    size_t n; long long total = 0, samples = 0;
    while ((n = std::fread(in.data(), 8, call, f)) > 0) {
      int consumed = 0;
      const int got = rx->general_work((int)out.size(), (int)n, in.data(), out.data(), consumed);
      if (got > 0) { std::fwrite(out.data(), 1, (size_t)got, o); total += got; }
      samples += (long long)n;
    }
    rx->stop();
    long long got;
    while ((got = rx->drain(out.data(), out.size())) > 0) { std::fwrite(out.data(), 1, (size_t)got, o); total += got; }
    std::fclose(f); std::fclose(o);
    const dvbt_rx_stream_info i = rx->info();
    std::printf("%lld samples -> %lld TS bytes (status %d)\n", samples, total, i.status);
  } catch (const std::exception &e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
  return 0;
}
