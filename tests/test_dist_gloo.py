"""N>1 path on CPU: world_size 2, gloo. Covers segment sharding and the single gather of decoded
TS bytes to rank 0 (the only inter-rank exchange of the design)."""
import os
import sys
import subprocess
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from gr_dvbt_amd import multi
    from oracle import pyoracle as po
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shards = multi.split_superframes(5, world)
    assert shards == [(0, 3), (3, 2)]
    # every rank decodes its own segment (here with the CPU oracle standing in for the GPU chain)
    c = po.cfg(po.QAM16, po.C1_2, po.T2k)
    ts = po.make_ts(504 * (shards[rank][1] + 1), 100 + rank)
    iq = po.tx(c, ts, lead_in=500, tail=3 * c.N)
    out = po.rx(c, iq, want=("ts",))["ts"]
    cap = 1 << 20
    buf = torch.zeros(cap, dtype=torch.uint8)
    buf[:len(out)] = torch.from_numpy(out.copy())
    res = multi.gather_ts(buf, len(out), cap, dst=0)
    if rank == 0:
        assert len(res) == world
        assert bytes(res[0].numpy()) == bytes(out)
        # rank 1's bytes must be what rank 1 decoded: recompute here
        ts1 = po.make_ts(504 * (shards[1][1] + 1), 101)
        out1 = po.rx(c, po.tx(c, ts1, lead_in=500, tail=3 * c.N), want=("ts",))["ts"]
        assert bytes(res[1].numpy()) == bytes(out1) and len(out1) > 1000
        print("GATHER_OK", [len(r) for r in res])
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_gather_gloo(tmp_path):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(w)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0]
