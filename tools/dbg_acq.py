import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from oracle import pyoracle as po
import gr_dvbt_amd as g
for (const, cr, mode, guard, nsf, lead, seed) in [(0, 2, 0, 3, 4, 2299, 5), (0, 4, 0, 3, 3, 4856, 6), (0, 2, 0, 3, 4, 2299, 77), (0, 0, 0, 3, 3, 2299, 6)]:
    c = po.cfg(const, cr, mode, guard=guard)
    ibits = c.payload * c.m * c.k // c.n
    ts = po.make_ts((272 * ibits * nsf) // (204 * 8), seed)
    iq = po.tx(c, ts, lead_in=lead, tail=3 * c.N)
    o = po.rx(c, iq, want=("ts",))
    rx = g.Rx(const, cr, mode, max_samples=len(iq), guard=guard, taps=True)
    rep = rx.run(iq)
    cps = rx.tap(g.TAP_CP_START)
    print(f"const{const} cr{cr} gi{guard} lead{lead} N+cp {c.N + c.cp}: gpu status {rep.status} nsym {rep.n_symbols} cp0 {rep.cp_start0} first {rep.first_out_symbol} | oracle nacq {o['n_acquired']} first {o['first_out_symbol']}")
    print("   gpu cp_start[:6]", cps[:6], " oracle", o["cp_start"][:6], "resume", rep.resume_sample, "first_call", rep.first_call, "ts", rep.n_ts_bytes, o["ts"].size)
    rx.close()
