"""rocprofv3 kernel + memory-copy trace of the ten-block C++ driver (single thread / thread per block, registered buffers, 64 symbols per call): which kernels and copies make a block's call"""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
threads = sys.argv[1] if len(sys.argv) > 1 else "0"
c = po.cfg(po.QAM64, po.C7_8, po.T8k)
iq = po.stream_slice(c, 8, 77)
tmp = tempfile.mkdtemp(dir="/dev/shm")
fin = os.path.join(tmp, "bb.cf32"); iq.tofile(fin)
exe = os.path.join(ROOT, "gr_dvbt_amd", "host", "rx_blocks_bench")
out = os.path.join(ROOT, "gpurun_out", "prof_blocks_t" + threads)
cmd = ["rocprofv3", "--kernel-trace", "--memory-copy-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "r", "--", exe, "8k", "qam64", "7/8", fin, os.path.join(tmp, "o.ts"), "64", threads, "1"]
r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
print(r.stdout[-600:], r.stderr[-200:])
os.remove(fin)
