"""Debug helper: run oracle and GPU chain on the same loopback baseband, print per-tap agreement."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
import gr_dvbt_amd as g

def one(const, cr, mode, nsf, lead=1000, seed=11, chunk=0, snr=None, snr_param=30.0):
    c = po.cfg(const, cr, mode)
    ibits = c.payload * c.m * c.k // c.n
    npk = (272 * ibits * nsf) // (204 * 8)
    ts = po.make_ts(npk, seed)
    iq = po.tx(c, ts, lead_in=lead, tail=3 * c.N)
    if snr is not None:
        rng = np.random.RandomState(5)
        p = np.mean(np.abs(iq[lead:lead + 100000]) ** 2)
        sig = np.sqrt(p / (10 ** (snr / 10)) / 2)
        iq = (iq + sig * (rng.randn(len(iq)) + 1j * rng.randn(len(iq)))).astype(np.complex64)
    t0 = time.time()
    o = po.rx(c, iq, snr_db=snr_param, want=("acq", "fft", "eq", "demap", "symdeint", "bitdeint", "vit", "deint", "rs", "ts"))
    t_or = time.time() - t0
    rx = g.Rx(const, cr, mode, max_samples=len(iq), taps=True, viterbi_chunk_bytes=chunk, snr_db=snr_param)
    t0 = time.time()
    rep = rx.run(iq)
    t_gpu = time.time() - t0
    print(f"== const{const} cr{cr} mode{mode} nsf{nsf} snr{snr}: oracle {t_or:.2f}s gpu(first call) {t_gpu:.3f}s")
    print("   report: status", rep.status, "nsym", rep.n_symbols, "first_out", rep.first_out_symbol, "nout", rep.n_out_symbols,
          "cp0", rep.cp_start0, "vit", rep.n_viterbi_bytes, "rs_items", rep.n_rs_items, "ts", rep.n_ts_bytes,
          "rs_fail", rep.rs_fail_words, "rs_corr", rep.rs_corrected_symbols)
    print("   oracle: nacq", o["n_acquired"], "first_out", o["first_out_symbol"], "vit", len(o["vit"]), "rs", len(o["rs"]), "ts", len(o["ts"]),
          "rs_fail", o["rs_fail"], "rs_corr", o["rs_corr"])
    ok = True
    cps = rx.tap(g.TAP_CP_START)
    n = min(len(cps), len(o["cp_start"]))
    print("   cp_start equal:", (cps[:n] == o["cp_start"][:n]).all(), len(cps), len(o["cp_start"]))
    si = rx.tap(g.TAP_SYMBOL_INDEX)
    n = min(len(si), len(o["sym_index"]) - 1)
    print("   sym_index equal:", (si[:n] == o["sym_index"][:n]).all(), len(si))
    for name, tap in (("acq", g.TAP_ACQ), ("fft", g.TAP_FFT), ("eq", g.TAP_EQ)):
        a = rx.tap(tap); b = o[name]
        n = min(len(a), len(b))
        if n == 0:
            print(f"   {name}: EMPTY gpu {a.shape} oracle {b.shape}"); ok = False; continue
        err = np.abs(a[:n] - b[:n]); scale = np.abs(b[:n]).max()
        print(f"   {name}: n {n} (gpu {len(a)} or {len(b)}) max abs err {err.max():.3e} rel-to-max {err.max()/scale:.3e}")
    for name, tap in (("demap", g.TAP_DEMAP), ("symdeint", g.TAP_SYMDEINT), ("bitdeint", g.TAP_BITDEINT), ("vit", g.TAP_VITERBI),
                      ("deint", g.TAP_DEINT), ("rs", g.TAP_RS), ("ts", g.TAP_TS)):
        a = rx.tap(tap); b = o[name]
        n = min(len(a), len(b))
        a2 = a[:n].reshape(-1); b2 = b[:n].reshape(-1)
        nd = int((a2 != b2).sum())
        first = int(np.argmax(a2 != b2)) if nd else -1
        print(f"   {name}: len gpu {a.size} oracle {b.size} diffs {nd} first@{first}")
        ok = ok and nd == 0 and a.size == b.size
    # timing of a second run
    rx.enable_timing(True)
    t0 = time.time(); rx.run(iq); t1 = time.time() - t0
    print("   second run wall %.4fs; stage ms:" % t1, {k: round(rx.stage_ms(k), 3) for k in ("acq", "fft", "demod", "inner", "viterbi", "rs", "total")},
          "Msps(dev)", len(iq) / rx.stage_ms("total") / 1e3)
    rx.close()
    return ok

if __name__ == "__main__":
    print("devices", g.device_count())
    r = [one(po.QAM16, po.C1_2, po.T2k, 3),
         one(po.QAM64, po.C7_8, po.T8k, 2),
         one(po.QPSK, po.C7_8, po.T8k, 3, snr=14.0, snr_param=14.0),
         one(po.QAM64, po.C7_8, po.T8k, 3, snr=22.0, snr_param=22.0),
         one(po.QAM16, po.C2_3, po.T2k, 4, snr=18.0, snr_param=18.0, lead=777),
         one(po.QAM64, po.C3_4, po.T2k, 3, chunk=128)]
    print("ALL OK" if all(r) else "MISMATCH")
