import sys; sys.path.insert(0, '.')
import numpy as np
import gr_dvbt_amd as g
from oracle import pyoracle as po
ppm = float(sys.argv[1]) if len(sys.argv) > 1 else -40.0
const, cr, mode, nsf = (g.QAM64, g.C7_8, g.T8k, 2) if len(sys.argv) > 2 else (g.QAM16, g.C1_2, g.T2k, 4)
c = po.cfg(const, cr, mode)
iq = po.clock_offset(po.stream_slice(c, nsf, 13), ppm)
o = po.rx(c, iq, want=("ts",))
rx = g.Rx(const, cr, mode, max_samples=len(iq))
rep = rx.run(iq)
cps = rx.tap(g.TAP_CP_START)
print("status", rep.status, "nsym", rep.n_symbols, o["n_acquired"], "cp0", rep.cp_start0, "call0", rep.first_call)
n = min(len(cps), len(o["cp_start"]))
d = np.flatnonzero(cps[:n] != o["cp_start"][:n])
print("first diffs", d[:10], "gpu", cps[d[:10]] if len(d) else "", "oracle", o["cp_start"][d[:10]] if len(d) else "")
print("gpu tail", cps[-5:], "oracle at that point", o["cp_start"][n - 5:n + 3])
