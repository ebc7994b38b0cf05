"""BASELINE config 5 at the prescribed noise (8k QPSK 7/8, 8 dB by default) through the synchronous entry, for rocprofv3 and for wall-clock timing:
`python tools/period_prof.py [snr_db] [superframes] [runs] [stream seed]` prints the time per run and the lock structure."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import pyoracle as po
import gr_dvbt_amd as g

snr = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
nsf = int(sys.argv[2]) if len(sys.argv) > 2 else 16
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 5
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 21
c = po.cfg(g.QPSK, g.C7_8, g.T8k)
iq = po.channel(po.stream_slice(c, nsf, seed), c.N, snr_db=snr, seed=5)
dev = torch.from_numpy(iq.view(np.float32)).cuda()
torch.cuda.synchronize()
rx = g.Rx(g.QPSK, g.C7_8, g.T8k, max_samples=len(iq), snr_db=snr)
rep = rx.run_device(dev.data_ptr(), len(iq))
t0 = time.perf_counter()
for _ in range(runs):
    rep = rx.run_device(dev.data_ptr(), len(iq))
dt = (time.perf_counter() - t0) / runs
per = rx.lock_periods()
print(json.dumps({"snr_db": snr, "superframes": nsf, "samples": int(len(iq)), "ms_per_run": round(dt * 1e3, 3), "msamples_per_s": round(len(iq) / dt / 1e6, 1),
                  "lock_periods_delivering": int(rep.n_lock_periods), "lock_periods": len(per), "symbols": int(rep.total_symbols),
                  "rs_fail_words": int(rep.rs_fail_words), "ts_bytes": int(rep.n_ts_bytes)}))
rx.close()
