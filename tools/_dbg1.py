import sys, os, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import numpy as np
from oracle import pyoracle as po
import gr_dvbt_amd as g
import snr_sweep
snr = float(sys.argv[1]) if len(sys.argv) > 1 else 9.0
c = po.cfg(po.QPSK, po.C7_8, po.T8k)
iq = po.channel(po.stream_slice(c, 3, snr_sweep.SEED_TS), c.N, snr_db=snr, seed=snr_sweep.SEED_NOISE)
o = po.rx(c, iq, snr_db=snr, want=("bitdeint", "vit"))
for verify in (1, -1):
    rx = g.Rx(po.QPSK, po.C7_8, po.T8k, max_samples=len(iq), snr_db=snr, taps=2, viterbi_verify=verify)
    rep = rx.run(iq)
    vit = rx.tap(g.TAP_VITERBI)
    periods = rx.period_taps()
    print("verify", verify, "periods", len(periods), rep.n_lock_periods, "vit bytes", rep.n_viterbi_bytes, len(o["vit"]), "lock periods", rx.lock_periods())
    po.lib().o_viterbi_decode.restype = C.c_size_t
    ob = o["bitdeint"].reshape(-1)
    pos = 0
    for bd, voff, vbytes in periods:
        bd = np.ascontiguousarray(bd)
        ref = np.zeros(bd.size * c.m * c.k // (8 * c.n) + 64, np.uint8)
        n = po.lib().o_viterbi_decode(C.byref(c), 768, bd.ctypes.data_as(C.c_void_p), C.c_size_t(bd.size), ref.ctypes.data_as(C.c_void_p))
        d = np.flatnonzero(vit[voff:voff + n] != ref[:n])
        od = np.flatnonzero(vit[voff:voff + n] != o["vit"][voff:voff + n])
        ib = np.flatnonzero(bd != ob[pos:pos + bd.size])
        print(" period: in", bd.size, bd.size / c.payload, "voff", voff, "vbytes", vbytes, "n", n, "diff vs own-input ref", d.size, (d[:3], d[-3:]) if d.size else "", "| vs oracle vit", od.size, od[:5], "| input bytes differing from oracle's", ib.size, ib[:5] // c.payload)
        pos += bd.size
    rx.close()
