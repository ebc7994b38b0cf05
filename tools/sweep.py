"""Randomised parity sweep (GPU box): random DVB-T parameters, segment lengths, lead-ins, Viterbi chunk sizes and noise levels;
clean cases must match the oracle at every integer tap, noisy ones after the RS decoder."""
import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (load torch's HIP runtime first)
from oracle import pyoracle as po
import gr_dvbt_amd as g

TAPS = (("demap", g.TAP_DEMAP), ("symdeint", g.TAP_SYMDEINT), ("bitdeint", g.TAP_BITDEINT), ("vit", g.TAP_VITERBI),
        ("deint", g.TAP_DEINT), ("rs", g.TAP_RS), ("ts", g.TAP_TS))
MIN_SNR = {0: 14.0, 1: 19.0, 2: 25.0}


def one(rng, idx):
    const, cr, mode, guard = int(rng.randint(0, 3)), int(rng.randint(0, 5)), int(rng.randint(0, 2)), int(rng.randint(0, 4))
    nsf = int(rng.randint(2, 4)) if mode == 1 else int(rng.randint(2, 6))
    lead, seed = int(rng.randint(0, 5000)), int(rng.randint(1, 1 << 30))
    chunk = int(rng.choice([0, 0, 64, 100, 333, 768, 1500, 4096]))
    noisy = rng.rand() < 0.4
    if os.environ.get("SWEEP_CHUNK"): chunk = int(os.environ["SWEEP_CHUNK"])
    if os.environ.get("SWEEP_CLEAN"): noisy = False
    c = po.cfg(const, cr, mode, guard=guard)
    ibits = c.payload * c.m * c.k // c.n
    ts = po.make_ts((272 * ibits * nsf) // (204 * 8), seed & 0xffff)
    iq = po.tx(c, ts, lead_in=lead, tail=3 * c.N)
    snr = None
    if noisy:
        snr = MIN_SNR[const] + 4 * rng.rand() + (3 if cr >= 3 else 0)
        p = np.mean(np.abs(iq[lead:lead + 100000]) ** 2)
        sig = np.sqrt(p / (10 ** (snr / 10)) / 2)
        iq = (iq + sig * (rng.randn(len(iq)) + 1j * rng.randn(len(iq)))).astype(np.complex64)
    sp = 30.0 if snr is None else float(snr)
    o = po.rx(c, iq, snr_db=sp, want=tuple(t[0] for t in TAPS))
    rx = g.Rx(const, cr, mode, max_samples=len(iq), guard=guard, taps=True, viterbi_chunk_bytes=chunk, snr_db=sp)
    rep = rx.run(iq)
    same_lock = rep.n_symbols == o["n_acquired"] and rep.first_out_symbol == o["first_out_symbol"]
    same_lock = same_lock and (rx.tap(g.TAP_CP_START) == o["cp_start"]).all()   # False after a start-up restart (symbols are counted from the restart)
    ok = True
    bad = []
    for name, tap in (TAPS if snr is None else TAPS[-2:]):
        a, b = rx.tap(tap).reshape(-1), o[name].reshape(-1)
        if a.size != b.size or (a != b).any():
            bad.append(name)
    rx.close()
    print(f"[{idx}] const{const} cr{cr} mode{mode} gi{guard} nsf{nsf} lead{lead} chunk{chunk} snr{None if snr is None else round(snr, 1)}: "
          f"nsym {rep.n_symbols} first {rep.first_out_symbol} ts {rep.n_ts_bytes} rs_corr {rep.rs_corrected_symbols} -> {'OK' if ok and not bad else 'MISMATCH ' + str(bad)}{'' if same_lock else ' (restarted)'}")
    return ok and not bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
    res = [one(rng, i) for i in range(n)]
    print("ALL OK" if all(res) else f"{res.count(False)} MISMATCHES")
