"""How often the Viterbi stage's repair pass has work, by channel: the passes' counters (dvbt_rx_viterbi_proof_total: chunks of every decoder launch, chunks decoded again) over
BASELINE config 5 (8k QPSK 7/8 + AWGN, 14 .. 7 dB), the headline mode (8k QAM64 7/8) at 30 .. 16 dB, and a hierarchical transmission -- several noise seeds each, samples resident.
`python tools/repair_stats.py [superframes] [seeds]` prints one JSON line per case."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
import gr_dvbt_amd as g

nsf = int(sys.argv[1]) if len(sys.argv) > 1 else 6
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cases = [("8k QPSK 7/8 (config 5)", g.QPSK, g.C7_8, g.T8k, 0, (14.0, 10.0, 9.0, 8.0, 7.0)), ("8k QAM64 7/8 (headline mode)", g.QAM64, g.C7_8, g.T8k, 0, (30.0, 24.0, 22.0, 20.0, 18.0)),
         ("2k QAM64 alpha 2 2/3 (hierarchical, HP stream)", g.QAM64, g.C2_3, g.T2k, g.ALPHA2, (30.0,))]
for name, const, cr, mode, hier, snrs in cases:
    c = po.cfg(const, cr, mode, hierarchy=hier)
    if hier:
        ibits = c.payload * c.m * c.k // c.n
        clean = po.tx(c, po.make_ts((272 * ibits * nsf * 4) // (204 * 8), 5), lead_in=500, tail=3 * c.N)
    else:
        clean = po.stream_slice(c, nsf, 21)
    for snr in snrs:
        tot = {"chunks": 0, "decoded_again": 0, "sequential": 0}
        per, fail_words, words = 0, 0, 0
        for seed in range(seeds):
            iq = po.channel(clean, c.N, snr_db=snr, seed=100 + seed) if snr < 30.0 else clean
            rx = g.Rx(const, cr, mode, max_samples=len(iq), hierarchy=hier, snr_db=snr)
            rep = rx.run(iq)
            p = rx.viterbi_proof_total()
            for k in tot:
                tot[k] += p[k]
            per += rep.n_lock_periods; fail_words += rep.rs_fail_words; words += rep.n_rs_bytes // 188
            rx.close()
            if snr >= 30.0:
                break
        print(json.dumps({"case": name, "snr_db": snr, "superframes": nsf, "runs": 1 if snr >= 30.0 else seeds, "lock_periods_delivering": per, "rs_words": words, "rs_fail_words": fail_words,
                          **tot, "decoded_again_per_million_chunks": round(1e6 * tot["decoded_again"] / max(tot["chunks"], 1), 1)}), flush=True)
