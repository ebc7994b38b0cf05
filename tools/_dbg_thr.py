import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import pyoracle as po
import gr_dvbt_amd as g
from gr_dvbt_amd.flowgraph import RxFlowgraph
c = po.cfg(g.QAM64, g.C7_8, g.T8k)
iq = po.stream_slice(c, 3, 9)
ref = po.rx(c, iq, want=("ts", "vit", "rs", "bitdeint"))
for trial in range(3):
    fg = RxFlowgraph(g.QAM64, g.C7_8, g.T8k, len(iq), mode="host", call_symbols=64)
    ts = fg.run_threaded(iq)
    S = fg.stages
    names = [st.blk.name for st in S]
    outs = {}
    for k in (5, 6, 8, 9):
        outs[names[k]] = S[k].out[:S[k].produced * S[k].out_item].copy()
    print("trial", trial, "calls", [st.calls for st in S], "produced", [st.produced for st in S])
    for nm, key in (("bit_inner_deinterleaver", "bitdeint"), ("viterbi_decoder", "vit"), ("reed_solomon_dec", "rs"), ("energy_descramble", "ts")):
        a, b = outs[nm], ref[key].reshape(-1)
        n = min(len(a), len(b))
        bad = np.flatnonzero(a[:n] != b[:n])
        print("  ", nm, len(a), len(b), "first mismatch", (int(bad[0]), len(bad)) if len(bad) else None)
    fg.close()
