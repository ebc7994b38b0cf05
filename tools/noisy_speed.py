"""Does the front end keep its speed on noisy / multipath input?  The parallel tracker falls back to the sequential walk (acq_track_kernel) when its fixed-point
iteration does not converge or a window leaves the precomputed lags; this prints the time of a whole synchronous run (dvbt_rx_segment_run_device) per channel condition.
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
         ("cfo +3.37 + 25 dB", dict(cfo=3.37, snr_db=25.0)), ("cfo 0.2", dict(cfo=0.2)),
         ("clock +20 ppm", dict(ppm=20.0, rx_snr=20.0)), ("clock -50 ppm + cfo 0.2 + 25 dB", dict(ppm=-50.0, cfo=0.2, snr_db=25.0, rx_snr=20.0))]
# (ofdm_sym_acquisition's snr parameter at 20 dB where the symbol timing falls between samples: at the flowgraphs' 30 dB the reference's own peak detector
# drops the lock on an 8k stream then, tests/test_gpu_drift.py)
c = po.cfg(g.QAM64, g.C7_8, g.T8k)
clean = po.stream_slice(c, nsf, 21)
for name, kw in CASES:
    kw = dict(kw); ppm = kw.pop("ppm", 0.0); rx_snr = kw.pop("rx_snr", None)
    iq = po.clock_offset(clean, ppm) if ppm else clean
    iq = po.channel(iq, c.N, seed=5, **kw) if kw else iq
    dev = torch.from_numpy(iq.view(np.float32)).cuda()
    torch.cuda.synchronize()
    rx = g.Rx(g.QAM64, g.C7_8, g.T8k, max_samples=len(iq), snr_db=rx_snr if rx_snr is not None else kw.get("snr_db", 30.0))
    import time
    rep = rx.run_device(dev.data_ptr(), len(iq))                   # the synchronous entry: through every start-up transient and lock period
    t0 = time.perf_counter()
    for _ in range(3):
        rep = rx.run_device(dev.data_ptr(), len(iq))
    dt = (time.perf_counter() - t0) / 3
    print(json.dumps({"case": name, "samples": int(len(iq)), "status": int(rep.status), "lock_periods": int(rep.n_lock_periods), "symbols": int(rep.total_symbols),
                      "ms_per_run": round(dt * 1e3, 3), "msamples_per_s": round(len(iq) / dt / 1e6, 1),
                      "rs_corrected": int(rep.rs_corrected_symbols), "rs_fail_words": int(rep.rs_fail_words), "ts_bytes": int(rep.n_ts_bytes)}), flush=True)
    rx.close()
