"""The streaming entry's host-side follower of energy_descramble (csrc/dvbt_stream.inc::descr_walk, reached through the test hook dvbt_debug_descr_follow; no device
needed) against the oracle's restatement of lib/energy_descramble_impl.cc:108-174 (oracle/o_outer.c::o_energy_descramble): RS output with its NSYNC on a random
phase, sync bytes hit by errors, packets missing at junctions (the phase jumps: the descrambler drops calls, searches, locks again), stretches with no sync at all --
the calls the follower delivers, descrambled, are the oracle's TS byte for byte, whatever windows the items become visible in."""
import ctypes as C

import numpy as np
import pytest

import gr_dvbt_amd as g


def _stream(rng, nitems, kind):
    npk = nitems * 8
    body = rng.integers(0, 256, size=(npk + 64, 188), dtype=np.uint8)
    body[:, 0] = 0x47
    phase = int(rng.integers(0, 8))
    body[phase::8, 0] = 0xB8
    if kind in ("errors", "all"):
        hit = rng.random(len(body)) < 0.03                             # sync bytes hit by errors: an NSYNC missed, an ordinary packet that looks like one
        body[hit, 0] = rng.integers(0, 256, size=int(hit.sum()), dtype=np.uint8)
    rows = list(range(len(body)))
    if kind in ("junctions", "all"):
        for _ in range(int(rng.integers(1, 5))):                         # packets missing: the NSYNC phase jumps
            a = int(rng.integers(8, max(9, len(rows) - 40)))
            del rows[a:a + int(rng.integers(1, 15))]
    out = body[rows][:npk].copy()
    if kind == "all":
        a = int(rng.integers(0, max(1, npk - 64)))
        out[a:a + 40, 0] = 0x47                                          # a stretch with no NSYNC at all: two-item drops
    return np.ascontiguousarray(out).reshape(-1)


@pytest.mark.parametrize("kind", ["clean", "errors", "junctions", "all"])
def test_follower_equals_the_oracles_descrambler(po, kind):
    L = g.binding.lib()
    O = po.lib()
    O.o_energy_descramble_groups.restype = C.c_size_t
    O.o_energy_descramble_groups.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.dvbt_debug_descr_follow.restype = C.c_int64
    L.dvbt_debug_descr_follow.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng({"clean": 1, "errors": 2, "junctions": 3, "all": 4}[kind])
    total, several = 0, 0
    for case in range(40):
        nitems = int(rng.integers(4, 120))
        rs = _stream(rng, nitems, kind)
        ref = np.zeros(nitems * 1504, np.uint8)
        n_ref = O.o_energy_descramble(rs.ctypes.data, nitems, ref.ctypes.data)
        ref = ref[:n_ref]
        # the items become visible in random steps (a walk window at a time), in one step, and one item at a time
        for windows in (sorted(set(int(x) for x in rng.integers(0, nitems + 1, size=int(rng.integers(1, 9)))) | {nitems}), [nitems], list(range(1, nitems + 1))):
            w = np.asarray(windows, np.int64)
            runs = np.zeros(2 * (nitems + 2), np.int64)
            n = L.dvbt_debug_descr_follow(rs.ctypes.data, nitems, w.ctypes.data, len(w), runs.ctypes.data, nitems + 2)
            assert n >= 0, L.dvbt_last_error()
            parts = []
            for k in range(n):
                first, count = int(runs[2 * k]), int(runs[2 * k + 1])
                assert count % 16 == 0 and rs[first * 188] == 0xB8            # whole two-item calls, each run begins on an NSYNC
                src = np.ascontiguousarray(rs[first * 188:(first + count) * 188])
                out = np.zeros(count * 188, np.uint8)
                O.o_energy_descramble_groups(src.ctypes.data, count // 8, out.ctypes.data)
                parts.append(out)
            ts = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
            assert len(ts) == len(ref) and (ts == ref).all(), (kind, case, len(windows), len(ts), len(ref))
            total += len(ts); several += n > 1
    assert total > 40 * 1504 and (kind == "clean" or several > 10)       # the cases do deliver, and the disturbed ones in several runs
