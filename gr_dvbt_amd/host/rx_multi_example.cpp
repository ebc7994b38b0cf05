// rx_multi_example.cpp -- BASELINE config 4's host without Python: ONE baseband stream decoded by `world` processes, one per GPU, the decoded packets brought
// to rank 0 by the design's single exchange per step (dvbt_rx_stream_gather: grouped ncclSend / ncclRecv on device buffers = an RCCL gather over xGMI).
// Every process reads the same file (a flowgraph would fan the source out), pushes it into its own dvbt_rx_stream (rank / world: it copies and decodes only
// the pieces k % world == rank), and joins a gather every GATHER_EVERY work() calls; rank 0 orders the runs it receives by their packet index -- that is the
// TS of one chain over the whole stream -- and writes the file.  No RCCL headers: the communicator comes from the library (dvbt_rccl_unique_id on rank 0, the
// 128 bytes carried to the others through a file, dvbt_rccl_comm_create everywhere).
//   rx_multi_example <rank> <world> <id file> <2k|8k> <qpsk|qam16|qam64> <1/2|2/3|3/4|5/6|7/8> <baseband.cf32> <out.ts> [superframes per piece] [device]
// started once per rank (e.g. `for r in 0 1 ... ; do rx_multi_example $r 8 /tmp/id ... & done`); rank r uses device r unless told otherwise.
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
    dvbt_rx_stream *st = nullptr;
    check(dvbt_rx_stream_create(&p, &st));
    dvbt_dims d; check(dvbt_get_dims(con, DVBT_NH, cr, DVBT_G1_32, mode, &d));
    std::FILE *f = std::fopen(argv[7], "rb"); if (!f) { std::perror("open"); return 1; }
    const size_t call = (size_t)64 * (d.fft_length + d.cp_length);
    const int GATHER_EVERY = 8, SLOT_PACKETS = 4096;
    std::vector<float> in(2 * call);
    std::vector<unsigned char> got((size_t)world * SLOT_PACKETS * 188);
    std::vector<dvbt_gather_chunk> chunks((size_t)world);
    std::map<long long, std::vector<unsigned char>> runs;            // rank 0: first packet -> bytes
    long long calls = 0, steps = 0, samples = 0;
    int all_done = 0;
    auto step = [&]() {
      const long long n = dvbt_rx_stream_gather(st, comm, 0, SLOT_PACKETS, rank == 0 ? got.data() : nullptr, got.size(), rank == 0 ? chunks.data() : nullptr, &all_done);
      check((int)(n < 0 ? n : 0));
      steps++;
      if (rank == 0) for (int r = 0; r < world; r++) if (chunks[r].nbytes > 0)
        runs[chunks[r].first_packet].assign(got.begin() + chunks[r].offset, got.begin() + chunks[r].offset + chunks[r].nbytes);
    };
    size_t n;
    while ((n = std::fread(in.data(), 8, call, f)) > 0) {
      check(dvbt_rx_stream_push(st, in.data(), n));
      samples += (long long)n;
      if (++calls % GATHER_EVERY == 0) step();
    }
    std::fclose(f);
    check(dvbt_rx_stream_finish(st));
    while (!all_done) step();
    dvbt_rx_stream_info inf; check(dvbt_rx_stream_status(st, &inf));
    int rc = 0;
    if (rank == 0) {
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
