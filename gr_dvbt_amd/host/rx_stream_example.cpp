// rx_stream_example.cpp -- apps/dvbt_rx_demo*.grc from the multiply_const on, with the ten receive blocks replaced by the one block over the
// streaming entry of libdvbt_hip (gr::dvbt::rx_hip; here its GNU Radio-free mirror of dvbt_blocks.hpp): file_source -> rx_hip -> file_sink.
// The loop below is what GNU Radio's single-threaded executor does with one block (gnuradio-runtime/lib/block_executor.cc): offer the items the source
// has (here `call` OFDM symbols per read), ask forecast(), call general_work() with an output buffer of `out_items` bytes, write what it returns; when the
// source has ended and its buffer is empty a block that still asks for input is done, one that asks for none is called until it returns WORK_DONE.
// NOTHING is fetched behind the block's back: the TS file holds exactly what general_work() returned.
//   rx_stream_example <mode 2k|8k> <constellation qpsk|qam16|qam64> <rate 1/2|2/3|3/4|5/6|7/8> <baseband.cf32> <out.ts> [symbols per call] [output buffer bytes] [superframes per piece]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "dvbt_blocks.hpp"

using namespace gr::dvbt_amd;

int main(int argc, char **argv)
{
  if (argc < 6) { std::printf("usage: %s <2k|8k> <qpsk|qam16|qam64> <1/2|2/3|3/4|5/6|7/8> <baseband.cf32> <out.ts> [symbols per call] [output buffer bytes] [superframes per piece]\n", argv[0]); return 2; }
  try {
    const dvbt_transmission_mode_t mode = !std::strcmp(argv[1], "8k") ? DVBT_T8k : DVBT_T2k;
    const dvbt_constellation_t con = !std::strcmp(argv[2], "qpsk") ? DVBT_QPSK : !std::strcmp(argv[2], "qam16") ? DVBT_QAM16 : DVBT_QAM64;
    const char *rates[] = {"1/2", "2/3", "3/4", "5/6", "7/8"};
    int cr = 0; for (int i = 0; i < 5; i++) if (!std::strcmp(argv[3], rates[i])) cr = i;
    const int symbols = argc > 6 ? std::atoi(argv[6]) : 64;
    const size_t out_items = argc > 7 ? (size_t)std::atoll(argv[7]) : (size_t)1 << 22;
    const int seg_sf = argc > 8 ? std::atoi(argv[8]) : 4;
    rx_hip::sptr rx = rx_hip::make(con, DVBT_NH, (dvbt_code_rate_t)cr, DVBT_G1_32, mode, 30.0f, 768, seg_sf);
    std::FILE *f = std::fopen(argv[4], "rb"), *o = std::fopen(argv[5], "wb");
    if (!f || !o) { std::perror("open"); return 1; }
    dvbt_dims d; check(dvbt_get_dims(con, DVBT_NH, cr, DVBT_G1_32, mode, &d));
    const size_t call = (size_t)symbols * (d.fft_length + d.cp_length);          // items the source hands over per read
    std::vector<float> in(2 * call); std::vector<unsigned char> out(out_items / 188 * 188);   // set_output_multiple(188)
    size_t have = 0; bool source_done = false; long long total = 0, samples = 0, calls = 0, backpressure = 0;
    std::vector<int> req(1);
    for (;;) {
      if (have == 0 && !source_done) { have = std::fread(in.data(), 8, call, f); if (have == 0) source_done = true; }
      rx->input_ended(source_done && have == 0);
      rx->forecast((int)out.size(), req);
      if ((size_t)req[0] > have) { if (source_done) break; else continue; }      // blocked on input whose upstream is done: the block is done
      int consumed = 0;
      const int got = rx->general_work((int)out.size(), (int)have, in.data(), out.data(), consumed);
      calls++;
      if (got == rx_hip::WORK_DONE) break;
      if (got > 0) { std::fwrite(out.data(), 1, (size_t)got, o); total += got; }
      if (consumed == 0 && have > 0) backpressure++;
      samples += consumed; have -= (size_t)consumed;                             // all or nothing
    }
    std::fclose(f); std::fclose(o);
    const dvbt_rx_stream_info i = rx->info();
    std::printf("%lld samples -> %lld TS bytes in %lld work calls (%lld under back-pressure), left inside %lld (status %d)\n", samples, total, calls, backpressure,
                (long long)i.ts_bytes_ready, i.status);
  } catch (const std::exception &e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
  return 0;
}
