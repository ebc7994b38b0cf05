"""rocprofv3 kernel trace of the C++ multi-GPU host's bench mode (one rank): where a piece's time goes on the device."""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1] if len(sys.argv) > 1 else "lent"
loops = sys.argv[2] if len(sys.argv) > 2 else "16"
c = po.cfg(po.QAM64, po.C7_8, po.T8k)
sf = 272 * (c.N + c.cp)
iq = po.stream_slice(c, 66, 77)
tmp = tempfile.mkdtemp(dir="/dev/shm")
fin, idf = os.path.join(tmp, "bb.cf32"), os.path.join(tmp, "nccl.id")
iq.tofile(fin)
exe = os.path.join(ROOT, "gr_dvbt_amd", "host", "rx_multi_example")
out = os.path.join(ROOT, "gpurun_out", "prof_cpp_" + mode)
cmd = ["rocprofv3", "--kernel-trace", "--memory-copy-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "r", "--", exe, "0", "1", idf, "8k", "qam64", "7/8", fin,
       os.path.join(tmp, "none.ts"), "64", "0", "bench", loops, str(po.STREAM_LEAD_IN + sf), str(64 * sf), str(8 * sf), "8", "400000"] + (["copy"] if mode == "copy" else [])
r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TMPDIR="/tmp"))
print(r.stdout[-400:], r.stderr[-300:])
for f in (fin, idf):
    if os.path.exists(f):
        os.remove(f)
