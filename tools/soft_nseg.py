"""The soft decoder's traceback segments (k_soft4.hpp S4_NSEG): fewer segments re-read fewer decisions from HBM ((S + 24) / S groups per group written) but make the
dependent chains longer; more segments the opposite.  Builds the library with 4 and with 8 (the product) segments, decodes 17 superframes of 8k QAM64 7/8 in soft mode with
each, checks the TS against the transmitted packets and prints the decoder's stage time.  `python tools/soft_nseg.py` on the GPU box."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
child = len(sys.argv) > 2 and sys.argv[1] == "--run"
if not child:
    for nseg in (4, 8):
        so = os.path.join(root, "tools", "_exp", f"libdvbt_nseg{nseg}.so")
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", f"-DS4_NSEG_N={nseg}",
                               "-o", so, os.path.join(root, "gr_dvbt_amd", "csrc", "dvbt_hip.hip")])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--run", so, str(nseg)])
    sys.exit(0)
import numpy as np
import torch
import gr_dvbt_amd.binding as b
b._SO = sys.argv[2]
import gr_dvbt_amd as g
from oracle import pyoracle as po
nseg, nsf = int(sys.argv[3]), 17
c = po.cfg(g.QAM64, g.C7_8, g.T8k)
iq = po.stream_slice(c, nsf, 21)
sent = {bytes(p) for p in po.stream_ts(c, 0, nsf, 21).reshape(-1, 188)}
dev = torch.from_numpy(iq.view(np.float32)).cuda()
rx = g.Rx(g.QAM64, g.C7_8, g.T8k, max_samples=len(iq), soft_decision=1)
rx.enable_timing(True)
for _ in range(6):
    rx.enqueue_device(dev.data_ptr(), len(iq)); rep = rx.finish()
ts = rx.tap(g.TAP_TS).reshape(-1, 188)
good = sum(1 for p in ts if bytes(p) in sent)
B = 304
S = ((256 // 8 + B - 30 + nseg * 6 - 1) // (nseg * 6)) * 6
print(json.dumps({"traceback_segments": nseg, "groups_per_segment_at_B_304": S, "decisions_read_per_written": round((S + 24) / S, 2), "viterbi_stage_ms": round(rx.stage_ms("viterbi"), 4),
                  "total_ms": round(rx.stage_ms("total"), 4), "ts_packets": int(len(ts)), "packets_not_transmitted": int(len(ts) - good)}))
rx.close()
