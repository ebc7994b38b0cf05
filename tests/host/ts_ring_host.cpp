// CPU check of dvbt::TsRing (gr_dvbt_amd/csrc/ts_ring.hpp), the placement of the streaming entry's decoded chunks in its pinned output ring: random
// sequences of placements and in-order releases on small rings; every placed chunk is filled with its own byte pattern and read back when it leaves, so an
// overlap of two live chunks, a chunk beyond the ring's end or a release that frees the wrong bytes shows as a corrupted pattern.
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <deque>
#include <vector>
#include "ts_ring.hpp"

struct Chunk { size_t off, len; bool in_ring; uint8_t tag; };

int main()
{
  srand(3);
  long placed = 0, in_ring = 0, wraps = 0, errors = 0;
  for (int trial = 0; trial < 400; trial++) {
    const size_t cap = 64 + rand() % 4000;
    dvbt::TsRing ring; ring.cap = cap;
    std::vector<uint8_t> buf(cap, 0);
    std::deque<Chunk> q;
    uint8_t tag = 1;
    size_t last_at = 0;
    auto release_front = [&]() {
      const Chunk c = q.front(); q.pop_front();
      if (c.in_ring) {
        for (size_t i = 0; i < c.len; i++) if (buf[c.off + i] != c.tag) { errors++; break; }
        size_t next = dvbt::TsRing::NONE;
        for (const Chunk &d : q) if (d.in_ring) { next = d.off; break; }
        ring.release(next);
      }
    };
    for (int op = 0; op < 3000; op++) {
      const int fill_bias = (op / 500) % 3;                       // phases: mostly placing, balanced, mostly releasing
      const bool do_place = q.empty() || rand() % 100 < (fill_bias == 0 ? 70 : fill_bias == 1 ? 50 : 30);
      if (do_place) {
        size_t len = 1 + rand() % (cap / 3 + 1);
        if (rand() % 50 == 0) len = cap + 1 + rand() % 10;         // larger than the ring: never placed
        if (rand() % 60 == 0) len = cap;                           // exactly the ring: only when it is empty
        bool live = false; for (const Chunk &d : q) live |= d.in_ring;
        const size_t at = ring.place(len);
        placed++;
        if (at == dvbt::TsRing::NONE) {
          if (!live && len <= cap) errors++;                       // an empty ring takes anything that fits
          q.push_back({0, len, false, 0});
        } else {
          if (at + len > cap) { errors++; continue; }
          for (const Chunk &d : q) if (d.in_ring && at < d.off + d.len && d.off < at + len) errors++;   // overlaps a live chunk
          if (live && at < last_at) wraps++;
          last_at = at;
          for (size_t i = 0; i < len; i++) buf[at + i] = tag;
          q.push_back({at, len, true, tag});
          tag = (uint8_t)(tag == 255 ? 1 : tag + 1);
          in_ring++;
        }
      } else release_front();
    }
    while (!q.empty()) release_front();
    if (!ring.empty) errors++;
  }
  printf("%ld placements, %ld in the ring, %ld wrap-arounds, %ld errors\n", placed, in_ring, wraps, errors);
  return errors != 0;
}
