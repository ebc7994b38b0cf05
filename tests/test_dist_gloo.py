"""N>1 path on CPU: world_size 2, gloo.  ONE stream is cut with multi.plan_cuts, every rank decodes its piece (here with
the oracle's cut mode standing in for the HIP chain -- there is no GPU in this container; tests/test_gpu_cut.py runs
the same plan_cuts / stitch_ts over the product), the pieces travel to rank 0 in the design's single collective
(multi.gather_ts: counts in the header of the same buffer) and rank 0 stitches them and compares with the TS of one
chain over the whole stream."""
import os
import sys
import subprocess
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    import gr_dvbt_amd as g
    from gr_dvbt_amd import multi
    from oracle import pyoracle as po
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []
    _gather = dist.gather
    def counting_gather(*a, **k):
        calls.append(1)
        return _gather(*a, **k)
    dist.gather = counting_gather
    const, cr, mode, nsf, seed = g.QAM16, g.C1_2, g.T2k, 6, 42
    c = po.cfg(const, cr, mode)
    d = g.get_dims(const, cr, mode)
    n = po.stream_len(c, nsf)
    # pre-scan on rank 0 (the head of the stream), broadcast of the two numbers every rank plans with
    plan = torch.zeros(2, dtype=torch.int64)
    if rank == 0:
        head = po.stream_slice(c, nsf, seed, 0, po.STREAM_LEAD_IN + 360 * (c.N + c.cp))
        r = po.rx(c, head, want=())
        plan[0], plan[1] = 0, r["first_out_symbol"]
    dist.broadcast(plan, src=0)
    cuts = multi.plan_cuts(d, n, int(plan[0]), int(plan[1]), world)
    assert [cu["count"] for cu in cuts] == [3, 2]
    cu = cuts[rank]
    iq = po.stream_slice(c, nsf, seed, cu["begin"], cu["end"])          # only this rank's piece of THE stream
    out = po.rx(c, iq, want=("ts",), sym_off=cu["sym_off"])
    out["n_ts_bytes"] = len(out["ts"]); out["status"] = 0
    meta = multi.piece_meta(out)
    cap = 1 << 20
    res = multi.gather_ts(torch.from_numpy(out["ts"].copy()), meta, cap, dst=0)
    assert len(calls) == 1, "the exchange step is ONE collective"
    if rank == 0:
        assert len(res) == world and res[1][0]["stream_symbol_offset"] == cuts[1]["sym_off"]
        st = multi.stitch_ts([(m, t.numpy()) for m, t in res], d)
        whole = po.rx(c, po.stream_slice(c, nsf, seed), want=("ts",))["ts"]
        assert len(st) == len(whole) > 100000 and (st == whole).all()
        print("STITCH_OK", [len(t) for _, t in res], len(st))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_cut_gather_stitch_gloo(tmp_path):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(w)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "STITCH_OK" in outs[0]
