// rx_multi_example.cpp -- BASELINE config 4's host without Python: ONE baseband stream decoded by `world` processes, one per GPU, the decoded packets brought
// to rank 0 by the design's single exchange per step (dvbt_rx_stream_gather_enqueue / _wait: ONE group of ncclSend / ncclRecv on device buffers = an RCCL gather
// over xGMI, asynchronous and double-buffered: step k + 1 is issued before step k's packets are consumed, the packets stay in device memory until the root
// downloads them: dvbt_rx_stream_set_device_output).
// Every process reads the same file (a flowgraph would fan the source out), pushes it into its own dvbt_rx_stream (rank / world: it copies and decodes only
// the pieces k % world == rank), and joins a step every GATHER_EVERY work() calls; rank 0 orders the runs it receives by their packet index -- that is the
// TS of one chain over the whole stream -- and writes the file.  No RCCL headers: the communicator comes from the library (dvbt_rccl_unique_id on rank 0, the
// 128 bytes carried to the others through a file, dvbt_rccl_comm_create everywhere).
//   rx_multi_example <rank> <world> <id file> <2k|8k> <qpsk|qam16|qam64> <1/2|2/3|3/4|5/6|7/8> <baseband.cf32> <out.ts> [superframes per piece] [device]
//                    [bench <loops> <loop_from> <loop_len> <samples per push> [pushes per exchange step] [slot packets] [copy] [mirror] [chains=N]]
// started once per rank (e.g. `for r in 0 1 ... ; do rx_multi_example $r 8 /tmp/id ... & done`); rank r uses device r unless told otherwise.
// bench: the throughput of this host on samples that are RESIDENT in device memory (what bench.py's line measures for the Python host): the file is uploaded
// once, then pushed from device memory (dvbt_rx_stream_push_device) -- its first loop_from + loop_len samples, then the stretch [loop_from, loop_from + loop_len)
// `loops` - 1 more times (a whole number of superframes from a superframe start on: the stream goes on seamlessly but for the encoder's and interleaver's
// memory at the seam); the samples are LENT to the stream (dvbt_rx_stream_params.borrow_device_pushes: a piece that lies in one stretch of the buffer is decoded in place; "copy"
// as the last argument: the default contract, every push copies); Msamples/s from the first push to the last packet at rank 0; out.ts is not written.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <vector>
#include "dvbt_blocks.hpp"

using namespace gr::dvbt_amd;

int main(int argc, char **argv)
{
  if (argc < 9) { std::printf("usage: %s <rank> <world> <id file> <2k|8k> <qpsk|qam16|qam64> <rate> <baseband.cf32> <out.ts> [superframes per piece] [device]\n", argv[0]); return 2; }
  try {
    const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
    const char *idfile = argv[3];
    const dvbt_transmission_mode_t mode = !std::strcmp(argv[4], "8k") ? DVBT_T8k : DVBT_T2k;
    const dvbt_constellation_t con = !std::strcmp(argv[5], "qpsk") ? DVBT_QPSK : !std::strcmp(argv[5], "qam16") ? DVBT_QAM16 : DVBT_QAM64;
    const char *rates[] = {"1/2", "2/3", "3/4", "5/6", "7/8"};
    int cr = 0; for (int i = 0; i < 5; i++) if (!std::strcmp(argv[6], rates[i])) cr = i;
    const int seg_sf = argc > 9 ? std::atoi(argv[9]) : 2;
    const int device = argc > 10 ? std::atoi(argv[10]) : rank;
    // ---- the communicator: rank 0 makes the id, the others wait for the file
    unsigned char id[128];
    if (rank == 0) {
      check(dvbt_rccl_unique_id(id));
      std::string tmp = std::string(idfile) + ".tmp";
      std::FILE *f = std::fopen(tmp.c_str(), "wb"); if (!f) { std::perror("id file"); return 1; }
      std::fwrite(id, 1, sizeof id, f); std::fclose(f); std::rename(tmp.c_str(), idfile);
    } else {
      for (int tries = 0;; tries++) {
        std::FILE *f = std::fopen(idfile, "rb");
        if (f) { const size_t n = std::fread(id, 1, sizeof id, f); std::fclose(f); if (n == sizeof id) break; }
        if (tries > 600) { std::fprintf(stderr, "rank %d: no id file\n", rank); return 1; }
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
      }
    }
    dvbt_rccl_comm *comm = nullptr;
    check(dvbt_rccl_comm_create(id, rank, world, device, &comm));
    // ---- this rank's stream
    dvbt_rx_stream_params p{};
    p.rx.constellation = con; p.rx.code_rate = cr; p.rx.transmission_mode = mode; p.rx.snr_db = 30.0f; p.rx.viterbi_bsize = 768; p.rx.descramble = 1; p.rx.device = device;
    p.segment_superframes = seg_sf; p.rank = rank; p.world = world;
    const bool bench_mode = argc > 15 && !std::strcmp(argv[11], "bench");
    p.borrow_device_pushes = bench_mode && !(argc > 18 && !std::strcmp(argv[18], "copy")) ? 1 : 0;   // bench: the resident samples are lent to the stream, not copied ("copy": the default contract)
    for (int i = 16; i < argc; i++) if (!std::strncmp(argv[i], "chains=", 7)) p.chains = std::atoi(argv[i] + 7);   // bench: chains of the stream object (dvbt_rx_stream_params.chains: 2 .. 4)
    dvbt_rx_stream *st = nullptr;
    check(dvbt_rx_stream_create(&p, &st));
    dvbt_dims d; check(dvbt_get_dims(con, DVBT_NH, cr, DVBT_G1_32, mode, &d));
    check(dvbt_rx_stream_set_device_output(st, 0));                    // the decoded TS waits in device memory for the exchange step
    const bool bench = argc > 15 && !std::strcmp(argv[11], "bench");
    // bench: the runs stay in rank 0's device memory (north_star: the clock ends at "last TS byte resident on rank 0") unless "mirror" asks for the page-locked copy as well
    bool mirror = false; for (int i = 16; i < argc; i++) if (!std::strcmp(argv[i], "mirror")) mirror = true;
    const int gflags = bench && !mirror ? DVBT_GATHER_DEVICE : 0;
    std::FILE *f = std::fopen(argv[7], "rb"); if (!f) { std::perror("open"); return 1; }
    const size_t call = (size_t)64 * (d.fft_length + d.cp_length);
    const int GATHER_EVERY = 8, SLOT_PACKETS = bench ? (argc > 17 ? std::atoi(argv[17]) : 1 << 16) : 4096;
    const long long pushes_per_step = bench && argc > 16 ? std::max(1, std::atoi(argv[16])) : 1;
    std::vector<float> in(2 * call);
    std::vector<unsigned char> got((size_t)world * SLOT_PACKETS * 188);
    std::vector<dvbt_gather_chunk> chunks((size_t)world);
    std::map<long long, std::vector<unsigned char>> runs;            // rank 0: first packet -> bytes
    long long calls = 0, steps = 0, samples = 0, ts_bytes = 0, order_errors = 0, last_end = -1;
    int all_done = 0, in_flight = 0;
    auto take = [&]() {                                              // the oldest step in flight
      // (bench: the runs are looked at where the step's download put them, dvbt_rccl_step_buffer: no copy on the host)
      const long long n = dvbt_rx_stream_gather_wait(st, comm, rank == 0 && !bench ? got.data() : nullptr, got.size(), rank == 0 ? chunks.data() : nullptr, &all_done);
      check((int)(n < 0 ? n : 0));
      in_flight--;
      if (rank == 0) for (int r = 0; r < world; r++) if (chunks[r].nbytes > 0) {
        ts_bytes += chunks[r].nbytes;
        if (bench) { if (world == 1 && last_end >= 0 && chunks[r].first_packet != last_end) order_errors++; last_end = chunks[r].first_packet + chunks[r].nbytes / 188; }
        else runs[chunks[r].first_packet].assign(got.begin() + chunks[r].offset, got.begin() + chunks[r].offset + chunks[r].nbytes);
      }
    };
    auto step = [&]() {                                              // issue a step; consume the one before it while this one travels
      check(dvbt_rx_stream_gather_enqueue_ex(st, comm, 0, SLOT_PACKETS, gflags));
      in_flight++; steps++;
      if (in_flight == 2) take();
    };
    double seconds = 0.0;
    if (bench) {
      const long long loops = std::atoll(argv[12]), loop_from = std::atoll(argv[13]), loop_len = std::atoll(argv[14]); const size_t per_push = (size_t)std::atoll(argv[15]);
      std::fseek(f, 0, SEEK_END); const size_t total = (size_t)std::ftell(f) / 8; std::fseek(f, 0, SEEK_SET);
      if ((size_t)(loop_from + loop_len) > total || loops < 1 || per_push < call) { std::fprintf(stderr, "bench: bad loop arguments\n"); return 2; }
      std::vector<float> all(2 * total);
      if (std::fread(all.data(), 8, total, f) != total) { std::perror("read"); return 1; }
      // lent samples: the looped stretch lies RING times back to back behind the stream's head, the pushes walk through that ring -- as a receiver's resident ring of
      // segments does: three of four seams between two passes are contiguous memory (their pieces are decoded where they lie), the fourth is the ring's wrap (gathered)
      const long long RING = p.borrow_device_pushes ? 4 : 1;
      const size_t dev_samples = p.borrow_device_pushes ? (size_t)(loop_from + RING * loop_len) : total;
      void *dev = dvbt_device_malloc(dev_samples * 8); if (!dev) { std::fprintf(stderr, "device allocation failed\n"); return 1; }
      check(dvbt_copy_to_device(dev, all.data(), std::min(total, dev_samples) * 8));
      for (long long k = 1; k < RING; k++) check(dvbt_copy_to_device((char *)dev + 8 * (size_t)(loop_from + k * loop_len), all.data() + 2 * (size_t)loop_from, (size_t)loop_len * 8));
      {   // warm-up: one pass of a stream of its own (the exchange buffers, the ring, the kernels' code are in place when the clock starts)
        dvbt_rx_stream *w = nullptr; check(dvbt_rx_stream_create(&p, &w)); check(dvbt_rx_stream_set_device_output(w, 0));
        check(dvbt_rx_stream_push_device(w, dev, (size_t)(loop_from + loop_len), nullptr)); check(dvbt_rx_stream_finish(w));
        int wd = 0, fl = 0;
        check(dvbt_rccl_comm_reserve(comm, 0, SLOT_PACKETS, gflags));   // (an allocation failure here, in front of the first step, not inside one)
        while (!wd) { check(dvbt_rx_stream_gather_enqueue_ex(w, comm, 0, SLOT_PACKETS, gflags)); fl++; const long long r = dvbt_rx_stream_gather_wait(w, comm, nullptr, 0, rank == 0 ? chunks.data() : nullptr, &wd); check((int)(r < 0 ? r : 0)); fl--; }
        dvbt_rx_stream_destroy(w);
      }
      const auto t0 = std::chrono::steady_clock::now();
      auto push_range = [&](size_t a, size_t e) {
        for (size_t at = a; at < e; at += per_push) {
          const size_t n = std::min(per_push, e - at);
          check(dvbt_rx_stream_push_device(st, (const char *)dev + 8 * at, n, nullptr));
          samples += (long long)n;
          if (++calls % pushes_per_step == 0) step();
        }
      };
      push_range(0, (size_t)(loop_from + loop_len));
      for (long long k = 1; k < loops; k++) { const size_t a = (size_t)(loop_from + (k % RING) * loop_len); push_range(a, a + (size_t)loop_len); }
      check(dvbt_rx_stream_finish(st));
      while (!all_done) { step(); }
      while (in_flight) take();
      seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      dvbt_device_free(dev);
    } else {
      size_t n;
      while ((n = std::fread(in.data(), 8, call, f)) > 0) {
        check(dvbt_rx_stream_push(st, in.data(), n));
        samples += (long long)n;
        if (++calls % GATHER_EVERY == 0) step();
      }
      check(dvbt_rx_stream_finish(st));
      while (!all_done) step();
      while (in_flight) take();
    }
    std::fclose(f);
    dvbt_rx_stream_info inf; check(dvbt_rx_stream_status(st, &inf));
    int rc = 0;
    if (rank == 0 && bench) {
      std::printf("{\"world\": %d, \"samples\": %lld, \"seconds\": %.4f, \"msamples_per_s\": %.1f, \"ts_bytes\": %lld, \"exchange_steps\": %lld, \"order_errors\": %lld, \"status\": %d, \"chains\": %d, \"runs\": \"%s\"}\n",
                  world, samples, seconds, samples / seconds / 1e6, ts_bytes, steps, order_errors, inf.status, p.chains ? p.chains : 2, mirror ? "page-locked mirror on rank 0 (exact download)" : "resident in rank 0's device memory");
      if (order_errors) rc = 1;
    } else if (rank == 0) {
      std::FILE *o = std::fopen(argv[8], "wb"); if (!o) { std::perror("open"); return 1; }
      long long at = -1, total = 0, gaps = 0;
      for (auto &kv : runs) {                                        // ordered by packet index: the single chain's TS
        if (at >= 0 && kv.first != at) gaps++;
        std::fwrite(kv.second.data(), 1, kv.second.size(), o); total += (long long)kv.second.size();
        at = kv.first + (long long)kv.second.size() / 188;
      }
      std::fclose(o);
      std::printf("world %d: %lld samples -> %lld TS bytes in %lld exchange steps, %zu runs, %lld gaps (status %d)\n", world, samples, total, steps, runs.size(), gaps, inf.status);
      if (gaps) rc = 1;
    }
    dvbt_rx_stream_destroy(st);
    dvbt_rccl_comm_destroy(comm);
    return rc;
  } catch (const std::exception &e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
}
