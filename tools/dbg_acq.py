import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from oracle import pyoracle as po
import gr_dvbt_amd as g
for (const, cr, mode, guard, nsf, lead, seed) in [(2, 1, 0, 1, 2, 4623, 5), (1, 3, 0, 2, 2, 3057, 6), (1, 0, 0, 0, 2, 4623, 7), (1, 0, 0, 0, 2, 2500, 7)]:
    c = po.cfg(const, cr, mode, guard=guard)
    ibits = c.payload * c.m * c.k // c.n
    ts = po.make_ts((272 * ibits * nsf) // (204 * 8), seed)
    iq = po.tx(c, ts, lead_in=lead, tail=3 * c.N)
    o = po.rx(c, iq, want=("ts",))
    rx = g.Rx(const, cr, mode, max_samples=len(iq), guard=guard, taps=True)
    rep = rx.run(iq)
    cps = rx.tap(g.TAP_CP_START)
    print(f"const{const} cr{cr} gi{guard} lead{lead} N+cp {c.N + c.cp}: gpu status {rep.status} nsym {rep.n_symbols} cp0 {rep.cp_start0} first {rep.first_out_symbol} | oracle nacq {o['n_acquired']} first {o['first_out_symbol']}")
    print("   gpu cp_start[:6]", cps[:6], " oracle", o["cp_start"][:6])
    rx.close()
