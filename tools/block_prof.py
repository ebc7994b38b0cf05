"""Where the time of the ten-block drop-in path goes (gr_dvbt_amd/flowgraph.py, BASELINE config 3, `call_symbols` OFDM symbols per call): seconds inside the
library's entry points per block (ctypes call included) against the whole run -- the rest is the Python driver that stands in for GNU Radio's scheduler."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from oracle import pyoracle as po
import gr_dvbt_amd as g
from gr_dvbt_amd.flowgraph import RxFlowgraph

cs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nsf = int(sys.argv[2]) if len(sys.argv) > 2 else 5
c = po.cfg(g.QAM64, g.C7_8, g.T8k)
iq = po.stream_slice(c, nsf, 77)
for mode in ("host", "device"):
    for rep in range(2):
        fg = RxFlowgraph(g.QAM64, g.C7_8, g.T8k, len(iq), mode=mode, call_symbols=cs)
        acc = {}
        for st in fg.stages:
            name = st.blk.name
            for fn in ("work", "work_device", "forecast"):
                orig = getattr(st.blk, fn)
                def wrap(*a, _o=orig, _k=(name, fn), **k):
                    t0 = time.perf_counter(); r = _o(*a, **k); acc[_k] = acc.get(_k, 0.0) + time.perf_counter() - t0; return r
                setattr(st.blk, fn, wrap)
        t0 = time.perf_counter()
        ts = fg.run(iq)
        dt = time.perf_counter() - t0
        calls = {st.blk.name: st.calls for st in fg.stages}
        fg.close()
    inside = sum(v for (n, f), v in acc.items() if f != "forecast")
    print(json.dumps({"mode": mode, "symbols_per_call": cs, "samples": int(len(iq)), "seconds": round(dt, 4), "msamples_per_s": round(len(iq) / dt / 1e6, 1),
                      "seconds_inside_work_calls": round(inside, 4), "seconds_in_forecast": round(sum(v for (n, f), v in acc.items() if f == "forecast"), 4),
                      "per_block_ms": {n: round(v * 1e3, 2) for (n, f), v in acc.items() if f != "forecast"}, "calls": calls}))
