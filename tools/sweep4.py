"""Randomised parity sweep over what round 4 added (GPU box): `python tools/sweep4.py [cases] [seed]`.  Per case one of
  hier   a loopback on an alpha = 1 / 2 / 4 constellation (16- / 64-QAM, any rate, mode, guard interval): demapper, symbol de-interleaver, both outputs of the bit
         de-interleaver and the decoder's bytes against the oracle;
  walk   a stream on which the reference's detector loses the CP lock again and again (QPSK / 16-QAM, any mode and guard interval, the noise level searched): the lock
         periods, the byte counts and the bytes behind the RS decoder against the oracle (the lock-period walk's one-launch tracker);
  auto   a stream decoded by a dvbt_rx_stream with constellation / hierarchy / code rate = AUTO against the oracle's chain with the true parameters;
  soft   the soft-decision path against the model (oracle/o_soft.c): soft values and decoded bytes identical (tests/test_gpu_soft.py::_model_check)."""
import os, sys, faulthandler
faulthandler.enable()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import torch  # noqa: F401
from oracle import pyoracle as po
import gr_dvbt_amd as g


def hier(rng):
    const, hr, cr, mode, guard = int(rng.randint(1, 3)), int(rng.randint(1, 4)), int(rng.randint(0, 5)), int(rng.randint(0, 2)), int(rng.randint(0, 4))
    c = po.cfg(const, cr, mode, guard=guard, hierarchy=hr)
    ibits = c.payload * c.m * c.k // c.n
    iq = po.tx(c, po.make_ts((272 * ibits * 2) // (204 * 8), int(rng.randint(1, 60000))), lead_in=int(rng.randint(0, 3000)), tail=3 * c.N)
    want = ("demap", "symdeint", "bitdeint", "bitdeint_lp", "vit")
    o = po.rx(c, iq, want=want)
    rx = g.Rx(const, cr, mode, max_samples=len(iq), guard=guard, hierarchy=hr, taps=True)
    rep = rx.run(iq)
    bad = [n for n, t in zip(want, (g.TAP_DEMAP, g.TAP_SYMDEINT, g.TAP_BITDEINT, g.TAP_BITDEINT_LP, g.TAP_VITERBI))
           if rx.tap(t).size != o[n].size or (rx.tap(t).reshape(-1) != o[n].reshape(-1)).any()]
    rx.close()
    return f"hier const{const} alpha{hr} cr{cr} mode{mode} gi{guard}: nsym {rep.n_symbols} first {rep.first_out_symbol} tps_hier {rep.tps_hierarchy}", not bad and rep.tps_hierarchy == hr and o["first_out_symbol"] >= 0


def walk(rng):
    const, mode, guard = int(rng.randint(0, 2)), int(rng.randint(0, 2)), int(rng.randint(0, 4))
    c = po.cfg(const, po.C1_2, mode, guard=guard)
    clean = po.stream_slice(c, 3 if mode == 1 else 6, int(rng.randint(1, 1000)))
    seed = int(rng.randint(1, 1000))
    for snr in (14.0, 12.0, 11.0, 10.0, 9.0, 8.0, 7.0, 6.0):
        iq = po.channel(clean, c.N, snr_db=snr + (5 if const else 0), seed=seed)
        o = po.rx(c, iq, snr_db=snr + (5 if const else 0), want=("vit", "rs", "ts"))
        if len(o["lock_periods"]) >= 2:
            break
    else:
        return f"walk const{const} mode{mode} gi{guard}: no noise level with repeated lock losses", None
    snr += 5 if const else 0
    rx = g.Rx(const, po.C1_2, mode, max_samples=len(iq), guard=guard, snr_db=snr)
    rep = rx.run(iq)
    L = c.N + c.cp
    got = [(off + fc * L, n) for (off, fc, cp0, n, fo) in rx.lock_periods() if n > 0]
    ok = got == o["lock_periods"] and rep.total_symbols == o["n_acquired"] and rep.n_viterbi_bytes == len(o["vit"]) and rep.n_rs_bytes == len(o["rs"]) and rep.n_ts_bytes == len(o["ts"])
    if ok and len(o["rs"]):
        ok = (rx.tap(g.TAP_RS) != o["rs"]).mean() < 1e-3
    rx.close()
    return f"walk const{const} mode{mode} gi{guard} snr{snr}: {len(got)} lock periods, {rep.total_symbols} symbols, {rep.n_ts_bytes} TS bytes", ok


def auto(rng):
    const, cr, mode, guard = int(rng.randint(0, 3)), int(rng.randint(0, 5)), int(rng.randint(0, 2)), int(rng.randint(0, 4))
    c = po.cfg(const, cr, mode, guard=guard)
    iq = po.stream_slice(c, 3 if mode == 1 else 6, int(rng.randint(1, 1000)))
    want = po.rx(c, iq, want=("ts",))["ts"]
    st = g.RxStream(g.AUTO, g.AUTO, mode, segment_superframes=int(rng.randint(1, 3)), guard=guard, hierarchy=g.AUTO)
    out, call = [], int(rng.randint(3000, 300000))
    for a in range(0, len(iq), call):
        st.push(iq[a:a + call]); out.append(st.pull())
    st.finish(); out.append(st.pull())
    info = st.info(); st.close()
    ts = np.concatenate(out)
    ok = (info.constellation, info.hierarchy, info.code_rate) == (const, 0, cr) and len(ts) == len(want) > 0 and (ts == want).all()
    return f"auto const{const} cr{cr} mode{mode} gi{guard} call{call}: detected {(info.constellation, info.hierarchy, info.code_rate)}, {len(ts)} TS bytes", ok


def soft(rng):
    import test_gpu_soft
    const, cr, mode = int(rng.randint(0, 3)), int(rng.randint(0, 5)), int(rng.randint(0, 2))
    snr = {0: 8.0, 1: 14.0, 2: 20.0}[const] + 1.5 * cr + 4 * rng.rand()
    rep = test_gpu_soft._model_check(po, const, cr, mode, 2 if mode == 1 else 4, float(snr), seed=int(rng.randint(1, 1000)), optional=True)
    return f"soft const{const} cr{cr} mode{mode} snr{snr:.1f}: " + ("first lock period too short" if rep is None else f"{rep.n_out_symbols} symbols, rs_corr {rep.rs_corrected_symbols}"), None if rep is None else True


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 77)
    res = []
    for i in range(n):
        kind = (hier, walk, auto, soft)[i % 4]
        try:
            msg, ok = kind(rng)
        except AssertionError as e:
            msg, ok = f"{kind.__name__}: assertion {str(e)[:200]}", False
        res.append(ok)
        print(f"[{i}] {msg} -> {'skipped' if ok is None else 'OK' if ok else 'MISMATCH'}", flush=True)
    print("ALL OK" if all(r is not False for r in res) else f"{res.count(False)} MISMATCHES", f"({res.count(True)} compared, {res.count(None)} skipped)")
