"""Does the front end keep its speed on noisy / multipath input?  The parallel tracker falls back to the sequential walk (acq_track_kernel) when its fixed-point
iteration does not converge or a window leaves the precomputed lags; this prints the acquisition stage's time and the whole step's per channel condition.
`python tools/noisy_speed.py [superframes]` on the GPU box, one JSON line per case."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import pyoracle as po
import gr_dvbt_amd as g

nsf = int(sys.argv[1]) if len(sys.argv) > 1 else 17
CASES = [("clean", {}), ("awgn 30 dB", dict(snr_db=30.0)), ("awgn 25 dB", dict(snr_db=25.0)), ("awgn 22 dB", dict(snr_db=22.0)),
         ("echo 0.3 cp -10 dB", dict(echoes=((77, 0.3),))), ("echo 0.9 cp -14 dB + 28 dB", dict(echoes=((230, 0.2),), snr_db=28.0)),
         ("cfo +3.37 + 25 dB", dict(cfo=3.37, snr_db=25.0)), ("cfo 0.2", dict(cfo=0.2))]
c = po.cfg(g.QAM64, g.C7_8, g.T8k)
clean = po.stream_slice(c, nsf, 21)
for name, kw in CASES:
    iq = po.channel(clean, c.N, seed=5, **kw) if kw else clean
    dev = torch.from_numpy(iq.view(np.float32)).cuda()
    torch.cuda.synchronize()
    rx = g.Rx(g.QAM64, g.C7_8, g.T8k, max_samples=len(iq), snr_db=kw.get("snr_db", 30.0))
    rx.enable_timing(True)
    for _ in range(3):
        rx.enqueue_device(dev.data_ptr(), len(iq)); rep = rx.finish()
    print(json.dumps({"case": name, "samples": int(len(iq)), "status": int(rep.status), "symbols": int(rep.n_symbols), "acq_stage_ms": round(rx.stage_ms("acq"), 3),
                      "total_ms": round(rx.stage_ms("total"), 3), "msamples_per_s": round(len(iq) / rx.stage_ms("total") / 1e3, 1),
                      "rs_corrected": int(rep.rs_corrected_symbols), "rs_fail_words": int(rep.rs_fail_words)}), flush=True)
    rx.close()
