"""The multi-GPU path's one collective on real hardware with one rank: bench.py --force-dist initialises torch.distributed with backend "nccl"
(= RCCL on ROCm), runs the double-buffered asynchronous gather of the decoded packets (device buffers, multi.gather_pieces) behind the next
step's decode, stitches what arrived and compares it with the transmitted packets (exit 1 on a wrong byte).  What an 8-GPU node adds to this
is N: the same code path with world > 1 (covered with two ranks under gloo in tests/test_dist_gloo.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(*args):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "4", "--warmup", "1", "--superframes", "8",
                        "--no-cpu-baseline", "--no-extras", *args], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [(), ("--segments", "2"), ("--pipeline", "1")], ids=["default", "two pieces per rank", "one step in flight"])
def test_rccl_gather_world_1(extra):
    d = _bench(*extra)
    c = d["config"]
    assert d["n_gpus"] == 1 and c["verified"] is True and c["wrong_bytes"] == 0 and c["ts_packets"] > 8 * 2000
    assert c["pieces_per_gpu"] == (2 if "--segments" in extra else 1)
    assert "one RCCL gather of TS per step" in c["parallelism"] and c["backend"] == "nccl" and c["ranks"] == 1
    assert d["roofline"]["frac"] > 0 and d["ms_per_step_dispersion"]["n"] >= 1


def _run(args, env=None, timeout=900):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=e, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_2_starts_two_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts the two ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), each builds its
    Job with world 2 (per-rank cuts of ONE stream, samples_owned, its own pieces), the pieces travel to rank 0 in one gather per step and rank 0 stitches
    and verifies them against the transmitted packets.  On the one-GPU test box both ranks decode on device 0 and the gather is staged through host
    tensors under gloo (RCCL refuses two ranks on one device); on an 8-GPU node the same code runs with backend nccl, one rank per GPU."""
    r = _run(["--gpus", "2", "--backend", "gloo", "--ranks-share-gpu", "--steps", "3", "--warmup", "1", "--superframes", "6", "--no-cpu-baseline", "--no-extras"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["ranks"] == 2 and c["backend"] == "gloo"
    assert c["stream_superframes"] == 13 and c["pieces_per_gpu"] == 1
    assert c["verified"] is True and c["wrong_bytes"] == 0 and c["ts_packets"] > 12 * 2000
    assert "cpu_baseline" not in d                      # rank 0 at N = 1 only
    assert d["value"] > 0 and d["scaling"] == "weak"


def test_bench_gpus_2_two_pieces_per_rank_one_step_in_flight():
    r = _run(["--gpus", "2", "--backend", "gloo", "--ranks-share-gpu", "--steps", "2", "--warmup", "1", "--superframes", "4", "--segments", "2", "--pipeline", "1",
              "--workload", "2k_qam16_1_2", "--no-cpu-baseline", "--no-extras"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["ranks"] == 2 and c["pieces_per_gpu"] == 2 and c["verified"] is True and c["wrong_bytes"] == 0


def test_bench_refuses_more_ranks_than_devices():
    import gr_dvbt_amd as g
    n = g.device_count()
    r = _run(["--gpus", str(n + 1), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras"], timeout=120)
    assert r.returncode != 0 and "HIP device(s) visible" in r.stderr and not r.stdout.strip()


def test_cpp_host_gathers_over_rccl(tmp_path):
    """gr_dvbt_amd/host/rx_multi_example: BASELINE config 4's host in C++ -- a sharded dvbt_rx_stream per process, the communicator from the library
    (dvbt_rccl_unique_id / dvbt_rccl_comm_create = ncclGetUniqueId / ncclCommInitRank through dlopen'd librccl), the packets brought to rank 0 by
    dvbt_rx_stream_gather_enqueue / _wait (ONE group of ncclSend / ncclRecv on device buffers per step, asynchronous and double-buffered, the packets device
    resident until the root's download), ordered by packet index and written out.
    One rank here (one GPU per test box); the TS file must be the oracle's chain over the whole stream."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import pyoracle as po
    import gr_dvbt_amd as g
    from conftest import host_example
    exe = host_example("rx_multi_example")
    c = po.cfg(g.QAM64, g.C7_8, g.T8k)
    iq = po.stream_slice(c, 9, 6)
    want = po.rx(c, iq, want=("ts",))["ts"]
    fin, fout, idf = tmp_path / "bb.cf32", tmp_path / "out.ts", tmp_path / "nccl.id"
    iq.tofile(fin)
    r = subprocess.run([exe, "0", "1", str(idf), "8k", "qam64", "7/8", str(fin), str(fout), "2"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    assert "0 gaps (status 0)" in r.stdout and "exchange steps" in r.stdout, r.stdout
    got = np.fromfile(fout, np.uint8)
    assert len(got) == len(want) > 0 and (got == want).all()


def test_cpp_host_bench_mode_on_resident_samples(tmp_path):
    """the same host on samples that are resident in device memory (what bench.py's cpp_multi_host line runs): the stretch behind the first superframe pushed three
    times from device memory, a step per push; every packet arrives in order and their number is what three passes deliver"""
    import json
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import pyoracle as po
    import gr_dvbt_amd as g
    from conftest import host_example
    exe = host_example("rx_multi_example")
    c = po.cfg(g.QAM64, g.C7_8, g.T8k)
    sf = 272 * (c.N + c.cp)
    nsf = 6
    iq = po.stream_slice(c, nsf, 6)
    fin, idf = tmp_path / "bb.cf32", tmp_path / "nccl.id"
    iq.tofile(fin)
    loops = 3
    r = subprocess.run([exe, "0", "1", str(idf), "8k", "qam64", "7/8", str(fin), str(tmp_path / "none.ts"), "2", "0", "bench", str(loops), str(po.STREAM_LEAD_IN + sf),
                        str(4 * sf), str(sf)], capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    pps = po.packets_per_superframe(c)
    assert d["order_errors"] == 0 and d["samples"] == po.STREAM_LEAD_IN + sf + loops * 4 * sf
    # the stream: 1 + 3 x 4 superframes, the reference's chain starts one frame early and holds two items back at the end
    assert abs(d["ts_bytes"] // 188 - (1 + loops * 4) * pps) <= pps, d


def test_cpp_host_device_resident_runs_and_exact_mirror(tmp_path):
    """the exchange step's two forms on rank 0: DVBT_GATHER_DEVICE leaves the runs in the root's device memory (nothing but the 64-byte headers is downloaded), the plain
    step mirrors them in page-locked memory -- per rank the header and the bytes the header declares, not the slot.  Driven through ctypes with one rank: both forms must
    deliver the same runs, which are the oracle's TS; the device-resident form refuses a host destination."""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import pyoracle as po
    import gr_dvbt_amd as g
    from gr_dvbt_amd import binding as b
    L = g.lib()
    c = po.cfg(g.QAM16, g.C1_2, g.T2k)
    iq = po.stream_slice(c, 7, 6)
    want = po.rx(c, iq, want=("ts",))["ts"]
    idb = (C.c_char * 128)()
    assert L.dvbt_rccl_unique_id(idb) == 0
    comm = C.c_void_p()
    L.dvbt_rccl_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    assert L.dvbt_rccl_comm_create(idb, 0, 1, 0, C.byref(comm)) == 0

    class Chunk(C.Structure):
        _fields_ = [("first_packet", C.c_int64), ("nbytes", C.c_int64), ("offset", C.c_int64)]
    L.dvbt_rx_stream_gather_enqueue_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.dvbt_rx_stream_gather_wait.restype = C.c_int64
    L.dvbt_rx_stream_gather_wait.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Chunk), C.POINTER(C.c_int)]
    L.dvbt_rccl_step_buffer.restype = C.c_void_p; L.dvbt_rccl_step_buffer.argtypes = [C.c_void_p]
    L.dvbt_rccl_step_device_buffer.restype = C.c_void_p; L.dvbt_rccl_step_device_buffer.argtypes = [C.c_void_p]
    L.dvbt_rccl_comm_reserve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.dvbt_rx_stream_set_device_output.argtypes = [C.c_void_p, C.c_size_t]
    L.dvbt_copy_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    SLOT = 3000
    outs = {}
    for flags in (1, 0):
        st = g.RxStream(g.QAM16, g.C1_2, g.T2k, segment_superframes=2, rank=0, world=1)
        assert L.dvbt_rx_stream_set_device_output(st.h, 0) == 0
        assert L.dvbt_rccl_comm_reserve(comm, 0, SLOT, flags) == 0
        runs, done, ch = {}, C.c_int(0), (Chunk * 1)()
        step = 40 * (c.N + c.cp)
        pos = 0
        while not done.value:
            if pos < len(iq):
                st.push(iq[pos:pos + step]); pos += step
                if pos >= len(iq):
                    st.finish()
            assert L.dvbt_rx_stream_gather_enqueue_ex(st.h, comm, 0, SLOT, flags) == 0, L.dvbt_last_error()
            if flags:
                assert L.dvbt_rx_stream_gather_wait(st.h, comm, (C.c_char * 16)(), 16, ch, C.byref(done)) < 0       # a device-resident step has no host destination (the step stays in flight)
            n = L.dvbt_rx_stream_gather_wait(st.h, comm, None, 0, ch, C.byref(done))
            assert n >= 0 and n == ch[0].nbytes <= SLOT * 188
            if n:
                buf = np.zeros(n, np.uint8)
                if flags:
                    assert L.dvbt_rccl_step_buffer(comm) is None
                    assert L.dvbt_copy_to_host(buf.ctypes.data_as(C.c_void_p), C.c_void_p(L.dvbt_rccl_step_device_buffer(comm) + ch[0].offset), n) == 0
                else:
                    C.memmove(buf.ctypes.data_as(C.c_void_p), C.c_void_p(L.dvbt_rccl_step_buffer(comm) + ch[0].offset), n)
                runs[ch[0].first_packet] = buf
        st.close()
        outs[flags] = np.concatenate([runs[k] for k in sorted(runs)])
    L.dvbt_rccl_comm_destroy.argtypes = [C.c_void_p]
    L.dvbt_rccl_comm_destroy(comm)
    assert len(outs[1]) == len(outs[0]) == len(want) > 0 and (outs[1] == want).all() and (outs[0] == want).all()


@pytest.mark.parametrize("WORLD", [2, 3])
def test_exchange_step_world_2_over_the_loopback_transport(WORLD):
    """The exchange step's world > 1 logic on the one GPU of a test box (RCCL refuses two ranks on a device): dvbt_rccl_comm_create_loopback -- the ranks are two threads, a send
    is a posted message, a receive copies device to device behind the sender's stream -- under the real dvbt_rx_stream_gather_enqueue_ex / _wait: two sharded streams pushed the
    same samples, every step one group (the slot to the root, the headers to both ranks), the root takes the runs from its device memory (DVBT_GATHER_DEVICE) or from the page-locked
    mirror (the OTHER rank's fill is read on the device by gather_download_kernel); ordered by packet index they are the oracle's TS.  Then a step in which rank 1's buffers are
    declared missing (dvbt_rccl_debug_fail_steps): its slot travels from the stream's sample buffer with the error flag, BOTH ranks' waits report the failure, nobody hangs."""
    import ctypes as C
    import threading
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import pyoracle as po
    import gr_dvbt_amd as g
    L = g.lib()
    c = po.cfg(g.QAM16, g.C1_2, g.T2k)
    iq = po.stream_slice(c, 9, 6)
    want = po.rx(c, iq, want=("ts",))["ts"]

    class Chunk(C.Structure):
        _fields_ = [("first_packet", C.c_int64), ("nbytes", C.c_int64), ("offset", C.c_int64)]
    L.dvbt_rccl_comm_create_loopback.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.dvbt_rccl_debug_fail_steps.argtypes = [C.c_void_p, C.c_int]
    L.dvbt_rx_stream_gather_enqueue_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.dvbt_rx_stream_gather_wait.restype = C.c_int64
    L.dvbt_rx_stream_gather_wait.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Chunk), C.POINTER(C.c_int)]
    L.dvbt_rccl_step_buffer.restype = C.c_void_p; L.dvbt_rccl_step_buffer.argtypes = [C.c_void_p]
    L.dvbt_rccl_step_device_buffer.restype = C.c_void_p; L.dvbt_rccl_step_device_buffer.argtypes = [C.c_void_p]
    L.dvbt_rx_stream_set_device_output.argtypes = [C.c_void_p, C.c_size_t]
    L.dvbt_copy_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.dvbt_rccl_comm_destroy.argtypes = [C.c_void_p]
    SLOT = 2500
    step = 48 * (c.N + c.cp)
    for flags, fail_at in ((1, None), (0, None), (1, 3)):
        group = C.c_void_p()
        comms = []
        for r in range(WORLD):
            cm = C.c_void_p()
            assert L.dvbt_rccl_comm_create_loopback(C.byref(group), r, WORLD, 0, C.byref(cm)) == 0, L.dvbt_last_error()
            comms.append(cm)
        runs, errors, failed_steps = {}, [], [0] * WORLD
        barrier = threading.Barrier(WORLD)

        def rank_main(r):
            try:
                st = g.RxStream(g.QAM16, g.C1_2, g.T2k, segment_superframes=2, rank=r, world=WORLD)
                assert L.dvbt_rx_stream_set_device_output(st.h, 0) == 0
                done, ch, pos, k = C.c_int(0), (Chunk * WORLD)(), 0, 0
                while not done.value:
                    if pos < len(iq):
                        st.push(iq[pos:pos + step]); pos += step
                        if pos >= len(iq):
                            st.finish()
                    k += 1
                    if fail_at is not None and r == 1 and k == fail_at:
                        assert L.dvbt_rccl_debug_fail_steps(comms[r], 1) == 0
                    assert L.dvbt_rx_stream_gather_enqueue_ex(st.h, comms[r], 0, SLOT, flags) == 0, L.dvbt_last_error()
                    n = L.dvbt_rx_stream_gather_wait(st.h, comms[r], None, 0, ch if r == 0 else None, C.byref(done))
                    if n < 0:
                        failed_steps[r] += 1
                        assert fail_at is not None and k == fail_at, (r, k, L.dvbt_last_error())
                        continue
                    if r == 0:
                        for q in range(WORLD):
                            nb = ch[q].nbytes
                            if nb:
                                buf = np.zeros(nb, np.uint8)
                                if flags:
                                    assert L.dvbt_copy_to_host(buf.ctypes.data_as(C.c_void_p), C.c_void_p(L.dvbt_rccl_step_device_buffer(comms[0]) + ch[q].offset), nb) == 0
                                else:
                                    C.memmove(buf.ctypes.data_as(C.c_void_p), C.c_void_p(L.dvbt_rccl_step_buffer(comms[0]) + ch[q].offset), nb)
                                runs[ch[q].first_packet] = (q, buf)
                barrier.wait(timeout=120)
                st.close()
            except BaseException as e:      # noqa: BLE001
                errors.append((r, repr(e)))
                try:
                    barrier.abort()
                except Exception:           # noqa: BLE001
                    pass
        ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(WORLD)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=300)
        assert not any(t.is_alive() for t in ths), "a rank hangs"
        assert not errors, errors
        for cm in comms:
            L.dvbt_rccl_comm_destroy(cm)
        ranks_seen = {q for q, _ in runs.values()}
        if fail_at is None:
            ts = np.concatenate([runs[k][1] for k in sorted(runs)])
            assert ranks_seen == set(range(WORLD)), ranks_seen                         # every rank contributed runs
            assert len(ts) == len(want) > 0 and (ts == want).all(), (flags, len(ts), len(want))
            assert failed_steps == [0] * WORLD
        else:
            assert failed_steps == [1] * WORLD, failed_steps                          # the failure of rank 1's step reached EVERY rank's wait; the streams went on
