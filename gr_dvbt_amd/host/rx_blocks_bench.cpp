// rx_blocks_bench.cpp -- apps/dvbt_rx_demo*.grc's ten receive blocks over the per-block C ABI (dvbt_<blk>_work, HOST buffers), driven by a thread-per-block
// scheduler written in C++: what GNU Radio's scheduler is (gnuradio-runtime: one thread per block, woken when a neighbour has produced or consumed; the executor
// halves the output request while the forecast cannot be met once the upstream is done).  gr_dvbt_amd/flowgraph.py is the same driver in Python (used by the
// parity tests); this one exists so that the drop-in path's throughput is not measured through an interpreter: every block call here is a C call from its own
// thread.  Tags travel with absolute item offsets; blocks with one output item per input item pass their input's tags on (GNU Radio's default propagation
// policy); the stock vector_to_stream between the bit de-interleaver and the Viterbi decoder is the factor `payload` on item counts and tag offsets.
// [registered 1]: the source and every block's output buffer are page-locked once (dvbt_host_register), as a GNU Radio shell would register its flowgraph
// buffers: the host-pointer entries then DMA straight from / to them instead of staging every item through pinned memory of the handle.
//   rx_blocks_bench <2k|8k> <qpsk|qam16|qam64> <1/2|2/3|3/4|5/6|7/8> <baseband.cf32> <out.ts> [symbols per call] [threads 0|1] [registered 0|1]
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#include "../../include/dvbt_hip.h"

struct Tag { long long off; int key, value; };
typedef int (*work_fn)(void *, int, int, const void *, void *, dvbt_sideband *);
typedef int (*forecast_fn)(const void *, int, int *);
typedef void (*destroy_fn)(void *);
struct Stage {
  void *h = nullptr; work_fn work = nullptr; forecast_fn forecast = nullptr; destroy_fn destroy = nullptr;
  size_t in_item = 1, out_item = 1; long long out_cap = 0;
  std::vector<unsigned char> out;
  long long r = 0, w = 0, produced = 0, calls = 0;      // items consumed / available (written by upstream) / produced
  double busy = 0.0;                                     // seconds spent inside work(): under a thread-per-block scheduler the run cannot be shorter than the busiest block's
  std::deque<Tag> tags;                                  // on the INPUT, absolute item offsets, ascending
  int per_call = 1, mult = 1;
};
static void chk(int r) { if (r < 0) throw std::runtime_error(std::string("libdvbt_hip: ") + dvbt_last_error()); }

int main(int argc, char **argv)
{
  if (argc < 6) { std::printf("usage: %s <2k|8k> <qpsk|qam16|qam64> <rate> <baseband.cf32> <out.ts> [symbols per call] [threads 0|1]\n", argv[0]); return 2; }
  try {
    const int mode = !std::strcmp(argv[1], "8k") ? DVBT_T8k : DVBT_T2k;
    const int con = !std::strcmp(argv[2], "qpsk") ? DVBT_QPSK : !std::strcmp(argv[2], "qam16") ? DVBT_QAM16 : DVBT_QAM64;
    const char *rates[] = {"1/2", "2/3", "3/4", "5/6", "7/8"};
    int cr = 0; for (int i = 0; i < 5; i++) if (!std::strcmp(argv[3], rates[i])) cr = i;
    const int cs = argc > 6 ? std::atoi(argv[6]) : 64;
    const bool threaded = argc > 7 ? std::atoi(argv[7]) != 0 : true;
    const bool registered = argc > 8 ? std::atoi(argv[8]) != 0 : false;
    dvbt_dims d; chk(dvbt_get_dims(con, DVBT_NH, cr, DVBT_G1_32, mode, &d));
    const int N = d.fft_length, cp = d.cp_length, P = d.payload_length, bsize = 768;
    std::FILE *f = std::fopen(argv[4], "rb"); if (!f) { std::perror("open"); return 1; }
    std::fseek(f, 0, SEEK_END); const long long nsamp = std::ftell(f) / 8; std::fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> src((size_t)nsamp * 8);
    if (std::fread(src.data(), 8, (size_t)nsamp, f) != (size_t)nsamp) { std::perror("read"); return 1; }
    std::fclose(f);
    const long long nsym = nsamp / (N + cp) + 2, nbytes = nsym * P * d.m * d.cr_k / (8 * d.cr_n) + 4096;
    const int vit_out_mult = bsize * d.cr_k / 8, vit_in_block = bsize * d.cr_n / d.m;
    std::vector<Stage> S(10);
    auto mk = [&](int k, void *h, work_fn w, forecast_fn fc, destroy_fn ds, size_t in_item, size_t out_item, long long out_cap, int per_call, int mult) {
      S[k].h = h; S[k].work = w; S[k].forecast = fc; S[k].destroy = ds; S[k].in_item = in_item; S[k].out_item = out_item; S[k].out_cap = out_cap;
      S[k].out.resize((size_t)out_cap * out_item + 64); S[k].per_call = per_call; S[k].mult = mult;
    };
#define BLK(k, name, params, ...) { dvbt_##name *h_ = nullptr; chk(dvbt_##name##_create(&(params), &h_)); \
    mk(k, h_, (work_fn)dvbt_##name##_work, (forecast_fn)dvbt_##name##_forecast, (destroy_fn)dvbt_##name##_destroy, __VA_ARGS__); }
    // parameters exactly as in apps/dvbt_rx_demo*.grc
    dvbt_ofdm_sym_acquisition_params pa = {1, N, d.Kmax + 1, cp, 30.0f};
    dvbt_fft_params pf = {N, 1, 1};
    dvbt_demod_reference_signals_params pd = {8, N, P, con, DVBT_NH, cr, cr, DVBT_G1_32, mode, 0, 0};
    dvbt_demap_params pm = {P, con, DVBT_NH, mode, 1.0f};
    dvbt_symbol_inner_interleaver_params ps = {P, mode, 0};
    dvbt_bit_inner_deinterleaver_params pb = {P, con, DVBT_NH, mode};
    dvbt_viterbi_decoder_params pv = {con, DVBT_NH, cr, bsize, 0, -1};
    dvbt_convolutional_deinterleaver_params pc = {136, 12, 17};
    dvbt_reed_solomon_dec_params pr = {2, 8, 0x11d, 255, 239, 8, 51, 8, 0};
    dvbt_energy_descramble_params pe = {8};
    const int ib = d.info_bits_per_symbol;
    const int vit_blocks = std::max(1, cs * P / vit_in_block);
    BLK(0, ofdm_sym_acquisition, pa, 8, (size_t)N * 8, nsym, cs, 1)
    BLK(1, fft, pf, (size_t)N * 8, (size_t)N * 8, nsym, cs, 1)
    BLK(2, demod_reference_signals, pd, (size_t)N * 8, (size_t)P * 8, nsym, cs, 1)
    BLK(3, demap, pm, (size_t)P * 8, (size_t)P, nsym, cs, 1)
    BLK(4, symbol_inner_interleaver, ps, (size_t)P, (size_t)P, nsym, cs, 1)
    BLK(5, bit_inner_deinterleaver, pb, (size_t)P, (size_t)P, nsym, cs, 1)
    BLK(6, viterbi_decoder, pv, 1, 1, nbytes, vit_blocks * vit_out_mult, vit_out_mult)
    BLK(7, convolutional_deinterleaver, pc, 1, 1632, nbytes / 1632 + 2, std::max(2, (cs * ib / 8 / 1632) & ~1), 2)
    BLK(8, reed_solomon_dec, pr, 1632, 1504, nbytes / 1632 + 2, std::max(2, cs * ib / 8 / 1632), 1)
    BLK(9, energy_descramble, pe, 1504, 1, nbytes, std::max(1, cs * ib / 8 / (4 * 1504)) * 4 * 1504, 4 * 1504)
    S[0].w = nsamp;
    if (registered) { chk(dvbt_host_register(src.data(), src.size())); for (auto &st : S) chk(dvbt_host_register(st.out.data(), st.out.size())); }
    std::mutex mu; std::condition_variable cv; std::vector<char> done(10, 0);
    // one general_work call of stage k if its input allows one; true when something was consumed or produced
    auto step = [&](int k, bool drain) -> bool {
      Stage &st = S[k];
      const unsigned char *in_buf = k == 0 ? src.data() : S[k - 1].out.data();
      long long avail; std::vector<dvbt_tag> tin;
      {
        std::lock_guard<std::mutex> g(mu);
        avail = st.w - st.r;
        for (const Tag &t : st.tags) if (t.off >= st.r) tin.push_back(dvbt_tag{t.off - st.r, t.key, t.value});
      }
      long long nout = std::min<long long>(st.per_call, st.out_cap - st.produced);
      if (nout <= 0 || avail <= 0) return false;
      int need = 0; chk(st.forecast(st.h, (int)nout, &need));
      while (drain && need > avail && nout / 2 >= st.mult) { nout = (nout / 2) / st.mult * st.mult; chk(st.forecast(st.h, (int)nout, &need)); }
      long long nin;
      if (k == 0) { if (avail < 2 * N + cp + 16) return false; nin = std::min<long long>(avail, (nout - 1) * (long long)(N + cp) + 2 * N + cp + 16); }
      else { nin = std::min<long long>(avail, need); if (k == 6 && nin < vit_in_block) return false; if (k == 7 && nin < 2 * 1632) return false; }
      size_t nt = 0; for (size_t i = 0; i < tin.size(); i++) if (tin[i].rel_offset < nin) tin[nt++] = tin[i];
      tin.resize(nt);
      std::vector<dvbt_tag> tout(4096);
      dvbt_sideband sb; sb.in_tags = tin.data(); sb.n_in_tags = (int)tin.size(); sb.out_tags = tout.data(); sb.out_cap = (int)tout.size(); sb.n_out_tags = 0; sb.n_consumed = 0;
      const auto w0 = std::chrono::steady_clock::now();
      const int produced = st.work(st.h, (int)nout, (int)nin, in_buf + (size_t)st.r * st.in_item, st.out.data() + (size_t)st.produced * st.out_item, &sb);
      st.busy += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
      chk(produced);
      st.calls++;
      const int consumed = sb.n_consumed;
      if (consumed == 0 && produced == 0) return false;
      std::lock_guard<std::mutex> g(mu);
      if (k + 1 < 10) {
        const long long scale = k == 5 ? P : 1;
        std::vector<Tag> add;
        for (int i = 0; i < sb.n_out_tags && i < sb.out_cap; i++) add.push_back(Tag{tout[i].rel_offset, tout[i].key, tout[i].value});
        if (k == 1 || k == 3 || k == 4 || k == 5 || k == 8) for (const dvbt_tag &t : tin) if (t.rel_offset < consumed) add.push_back(Tag{t.rel_offset, t.key, t.value});
        std::stable_sort(add.begin(), add.end(), [](const Tag &a, const Tag &b) { return a.off < b.off; });
        for (const Tag &t : add) S[k + 1].tags.push_back(Tag{(st.produced + t.off) * scale, t.key, t.value});
      }
      st.r += consumed;
      while (!st.tags.empty() && st.tags.front().off < st.r) st.tags.pop_front();
      st.produced += produced;
      if (k + 1 < 10) S[k + 1].w = st.produced * (k == 5 ? P : 1);
      return true;
    };
    const auto t0 = std::chrono::steady_clock::now();
    if (threaded) {
      std::vector<std::thread> th;
      std::string err;
      for (int k = 0; k < 10; k++) th.emplace_back([&, k]() {
        try {
          for (;;) {
            if (step(k, false)) { cv.notify_all(); continue; }
            bool up_done;
            { std::unique_lock<std::mutex> g(mu); up_done = k == 0 || done[k - 1]; if (!up_done) cv.wait_for(g, std::chrono::milliseconds(2)); }
            if (up_done) { if (!step(k, true)) break; cv.notify_all(); }
          }
        } catch (const std::exception &e) { std::lock_guard<std::mutex> g(mu); err = e.what(); }
        { std::lock_guard<std::mutex> g(mu); done[k] = 1; }
        cv.notify_all();
      });
      for (auto &t : th) t.join();
      if (!err.empty()) throw std::runtime_error(err);
    } else {
      for (int drain = 0; drain < 2; drain++)
        for (bool progress = true; progress;) { progress = false; for (int k = 0; k < 10; k++) while (step(k, drain != 0)) progress = true; }
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::FILE *o = std::fopen(argv[5], "wb"); if (!o) { std::perror("open"); return 1; }
    std::fwrite(S[9].out.data(), 1, (size_t)S[9].produced, o); std::fclose(o);
    long long calls = 0; for (auto &st : S) calls += st.calls;
    double busiest = 0.0, busy_sum = 0.0; int who = 0;
    for (int k = 0; k < 10; k++) { busy_sum += S[k].busy; if (S[k].busy > busiest) { busiest = S[k].busy; who = k; } }
    const char *names[] = {"ofdm_sym_acquisition", "fft", "demod_reference_signals", "dvbt_demap", "symbol_inner_interleaver", "bit_inner_deinterleaver", "viterbi_decoder",
                           "convolutional_deinterleaver", "reed_solomon_dec", "energy_descramble"};
    std::printf("{\"samples\": %lld, \"seconds\": %.5f, \"msamples_per_s\": %.2f, \"symbols_per_call\": %d, \"thread_per_block\": %s, \"registered_buffers\": %s, \"block_calls\": %lld, \"ts_bytes\": %lld, "
                "\"busiest_block\": \"%s\", \"busiest_block_seconds\": %.5f, \"busiest_block_bound_msamples_per_s\": %.2f, \"all_blocks_busy_seconds\": %.5f, \"ms_per_call_busiest\": %.4f}\n",
                nsamp, dt, nsamp / dt / 1e6, cs, threaded ? "true" : "false", registered ? "true" : "false", calls, S[9].produced,
                names[who], busiest, nsamp / busiest / 1e6, busy_sum, 1e3 * busiest / (double)std::max<long long>(S[who].calls, 1));
    if (registered) { dvbt_host_unregister(src.data()); for (auto &st : S) dvbt_host_unregister(st.out.data()); }
    for (auto &st : S) st.destroy(st.h);
  } catch (const std::exception &e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
  return 0;
}
