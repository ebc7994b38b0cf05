"""Lock periods of a channel case (tests/test_gpu_channel.py) on the GPU against the oracle's: prints the first divergence."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa: F401
from oracle import pyoracle as po
import gr_dvbt_amd as g
import test_gpu_channel as T

which = sys.argv[1] if len(sys.argv) > 1 else "2k echo 0.9 cp -6 dB"
case = next(c for grp in (T.CLEAN, T.NOISY, T.LOSSY) for c in grp if c[0] == which)
name, cfgt, chan = case[:3]
c, iq = T._stream(po, cfgt, **chan)
snr = chan.get("snr_db") or 30.0
o = po.rx(c, iq, snr_db=snr, want=("ts",))
rx = g.Rx(cfgt[0], cfgt[1], cfgt[2], max_samples=len(iq), snr_db=snr)
rep = rx.run(iq)
L = c.N + c.cp
gp = [(off + fc * L, n, off, fc, cp0) for (off, fc, cp0, n, fo) in rx.lock_periods()]
gpn = [(a, n) for (a, n, *_r) in gp if n > 0]
op = o["lock_periods"]
print("oracle periods", len(op), "symbols", o["n_acquired"], "| gpu periods", len(gp), "with symbols", len(gpn), "symbols", rep.total_symbols)
for i in range(max(len(op), len(gpn))):
    a = op[i] if i < len(op) else None
    b = gpn[i] if i < len(gpn) else None
    if a != b:
        print("first divergence at period", i, "oracle", op[max(0, i - 2):i + 3], "gpu", gpn[max(0, i - 2):i + 3])
        k = next(j for j, q in enumerate(gp) if (q[0], q[1]) == gpn[i]) if b else len(gp) - 1
        print("gpu raw periods around:", gp[max(0, k - 4):k + 3])
        break
else:
    print("identical")
