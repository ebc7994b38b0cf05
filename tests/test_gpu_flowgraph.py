"""The drop-in path: the ten receive blocks driven one general_work-sized call at a time through the per-block C ABI
(gr_dvbt_amd/flowgraph.py), with host buffers (dvbt_<blk>_work) and with device buffers (dvbt_<blk>_work_device).  The TS
must be the oracle's, whatever the call size."""
import numpy as np
import pytest

import gr_dvbt_amd as g
from gr_dvbt_amd.flowgraph import RxFlowgraph

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("const,cr,mode_t,nsf", [(g.QAM16, g.C1_2, g.T2k, 3), (g.QAM64, g.C7_8, g.T8k, 2)])
@pytest.mark.parametrize("mode,call_symbols", [("host", 4), ("device", 1), ("device", 4), ("device", 33)])
def test_block_by_block_equals_the_oracle(po, const, cr, mode_t, nsf, mode, call_symbols):
    c = po.cfg(const, cr, mode_t)
    iq = po.stream_slice(c, nsf, 9)
    ref = po.rx(c, iq, want=("ts",))["ts"]
    fg = RxFlowgraph(const, cr, mode_t, len(iq), mode=mode, call_symbols=call_symbols)
    ts = fg.run(iq)
    calls = [st.calls for st in fg.stages]
    fg.close()
    assert min(calls) > 0
    # the block-by-block chain stops a few items earlier or later than the whole-segment chain depending on where the calls' windows end
    # (the scheduler's leftovers); what it delivers must be the oracle's stream from its first byte on
    n = min(len(ts), len(ref))
    assert n > 0.9 * len(ref) and abs(len(ts) - len(ref)) <= 64 * 1504
    assert (ts[:n] == ref[:n]).all()


@pytest.mark.parametrize("const,cr,mode_t,nsf,call_symbols", [(g.QAM16, g.C1_2, g.T2k, 3, 16), (g.QAM64, g.C7_8, g.T8k, 3, 64)])
def test_thread_per_block_scheduler_equals_the_oracle(po, const, cr, mode_t, nsf, call_symbols):
    """the ten blocks under a thread-per-block scheduler (what GNU Radio's is): every block's general_work in its own thread, calls of whatever size its input
    allows at that moment, host buffers.  The TS must be the oracle's whatever the interleaving of the calls was."""
    c = po.cfg(const, cr, mode_t)
    iq = po.stream_slice(c, nsf, 9)
    ref = po.rx(c, iq, want=("ts",))["ts"]
    for _ in range(2):                                                     # two runs: two different interleavings
        fg = RxFlowgraph(const, cr, mode_t, len(iq), mode="host", call_symbols=call_symbols)
        ts = fg.run_threaded(iq)
        calls = [st.calls for st in fg.stages]
        fg.close()
        assert min(calls) > 0
        n = min(len(ts), len(ref))
        assert n > 0.9 * len(ref) and abs(len(ts) - len(ref)) <= 64 * 1504
        assert (ts[:n] == ref[:n]).all()


@pytest.mark.parametrize("threads", [1, 0], ids=["a thread per block", "one thread"])
def test_cpp_thread_per_block_driver_equals_the_oracle(po, tmp_path, threads):
    """gr_dvbt_amd/host/rx_blocks_bench: the ten blocks over dvbt_<blk>_work (host buffers) under a thread-per-block scheduler written in C++ (the form the
    drop-in path's throughput is measured in: no interpreter between the calls).  The TS file must be the oracle's."""
    import subprocess, json
    from conftest import host_example
    exe = host_example("rx_blocks_bench")
    const, cr, mode_t = g.QAM64, g.C7_8, g.T8k
    c = po.cfg(const, cr, mode_t)
    iq = po.stream_slice(c, 3, 9)
    ref = po.rx(c, iq, want=("ts",))["ts"]
    fin, fout = tmp_path / "bb.cf32", tmp_path / "out.ts"
    iq.tofile(fin)
    out = subprocess.check_output([exe, "8k", "qam64", "7/8", str(fin), str(fout), "64", str(threads)], text=True)
    info = json.loads(out.strip().splitlines()[-1])
    ts = np.fromfile(fout, np.uint8)
    n = min(len(ts), len(ref))
    assert info["samples"] == len(iq) and n > 0.9 * len(ref) and abs(len(ts) - len(ref)) <= 64 * 1504
    assert (ts[:n] == ref[:n]).all()


def test_registered_host_buffers(po, tmp_path):
    """dvbt_host_register: the flowgraph's buffers page-locked once; the host-pointer entries DMA straight from / to them (no staging).  Same TS -- from the Python
    driver and from the C++ thread-per-block driver."""
    import subprocess, json
    from conftest import host_example
    const, cr, mode_t = g.QAM64, g.C7_8, g.T8k
    c = po.cfg(const, cr, mode_t)
    iq = po.stream_slice(c, 3, 9)
    ref = po.rx(c, iq, want=("ts",))["ts"]
    fg = RxFlowgraph(const, cr, mode_t, len(iq), mode="host", call_symbols=64, register_buffers=True)
    ts = fg.run_threaded(iq)
    assert len(fg.registered) == 11                                   # the source and ten output buffers
    fg.close()
    n = min(len(ts), len(ref))
    assert n > 0.9 * len(ref) and (ts[:n] == ref[:n]).all()
    exe = host_example("rx_blocks_bench")
    fin, fout = tmp_path / "bb.cf32", tmp_path / "out.ts"
    iq.tofile(fin)
    out = subprocess.check_output([exe, "8k", "qam64", "7/8", str(fin), str(fout), "64", "1", "1"], text=True)
    assert json.loads(out.strip().splitlines()[-1])["registered_buffers"] is True
    ts = np.fromfile(fout, np.uint8)
    n = min(len(ts), len(ref))
    assert n > 0.9 * len(ref) and (ts[:n] == ref[:n]).all()
