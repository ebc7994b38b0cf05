// ts_ring.hpp -- where the streaming entry's decoded chunks go inside its pinned output ring (dvbt_stream.inc): variable-length chunks, placed contiguously,
// released in the order they were placed.  Host-only arithmetic, no HIP: tests/test_ts_ring_host.py compiles it with g++ and drives it with random sequences.
#pragma once
#include <cstddef>

namespace dvbt {

struct TsRing {
  size_t cap = 0, head = 0, tail = 0;      // live chunks lie in [tail, head) when head > tail; once the placements have turned around (head < tail) in
                                           // [tail, where they turned) and [0, head)
  bool empty = true;
  static constexpr size_t NONE = (size_t)-1;

  // offset of `len` contiguous bytes, or NONE when the ring has no room for them right now (the caller then keeps the chunk elsewhere)
  size_t place(size_t len)
  {
    if (len == 0 || len > cap) return NONE;
    size_t at = NONE;
    if (empty) { head = tail = 0; at = 0; }
    else if (head > tail) {
      if (head + len <= cap) at = head;                           // behind the newest chunk
      else if (len < tail) at = 0;                                // no room up to the end: from the ring's start, short of the oldest chunk
    } else if (head + len < tail) at = head;                      // wrapped: between the newest chunk and the oldest
    if (at == NONE) return NONE;
    head = at + len; empty = false;
    return at;
  }
  // the oldest chunk of the ring leaves; next_begin = offset of the next chunk still in the ring (NONE: it was the last one)
  void release(size_t next_begin)
  {
    if (next_begin == NONE) { empty = true; head = tail = 0; }
    else tail = next_begin;
  }
};

}  // namespace dvbt
