import sys; sys.path.insert(0, '.')
import numpy as np
import gr_dvbt_amd as g
from oracle import pyoracle as po
const, cr, mode, nsf, ppm, snr = g.QAM64, g.C3_4, g.T2k, 3, 55.0, 30.0
c = po.cfg(const, cr, mode)
iq = po.clock_offset(po.stream_slice(c, nsf, 13), ppm)
o = po.rx(c, iq, snr_db=snr, want=("ts",))
print("oracle nacq", o["n_acquired"], "cps", o["cp_start"][:6], "first_out", o["first_out_symbol"])
rx = g.Rx(const, cr, mode, max_samples=len(iq), snr_db=snr)
rep = rx.run(iq)
cps = rx.tap(g.TAP_CP_START)
print("gpu status", rep.status, "nsym", rep.n_symbols, "cp0", rep.cp_start0, "call0", rep.first_call, "seg_off", rep.segment_offset, "first_out", rep.first_out_symbol, "cps", cps[:6])
# run the GPU on the stream cut at the oracle's restart point to see what a fresh acquisition finds there
L = c.N + c.cp
for off in (L + L // 2,):
    rx2 = g.Rx(const, cr, mode, max_samples=len(iq), snr_db=snr)
    r2 = rx2.run(iq[off:])
    print("gpu fresh at", off, "status", r2.status, "cp0", r2.cp_start0, "call0", r2.first_call, "seg_off", r2.segment_offset, rx2.tap(g.TAP_CP_START)[:4])
    o2 = po.rx(c, iq[off:], snr_db=snr, want=())
    print("oracle fresh at", off, o2["cp_start"][:4], o2["n_acquired"])
