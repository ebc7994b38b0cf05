"""BASELINE config 5 where it means something: 8k QPSK 7/8 + AWGN, SNR swept from where the chain is error free down to where the
CP lock and the RS decoder give up ("fixed SNR chosen so pre-Viterbi BER ~ 1e-2" -- that is 8 dB here).  Per point, on the
IDENTICAL noisy samples: lock structure (symbols acquired, lock periods that delivered), pre-Viterbi BER, RS statistics, packet error
rate against the transmitted packets -- HIP path vs oracle -- and the time of the A8+A9 launch under that load.

ofdm_sym_acquisition's snr parameter is set to the channel's SNR (with the demo flowgraphs' constant 30 the reference's peak detector
never holds a lock below ~20 dB: rho ~ 1 makes lambda's noise floor comparable to its peak).

Used by tests/test_gpu_config5.py (assertions) and by hand (`python tools/snr_sweep.py [superframes]` prints one JSON line per point)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

SNRS = (14.0, 12.0, 10.0, 9.0, 8.0, 7.0, 5.0)
SEED_TS, SEED_NOISE = 21, 5


def packet_stats(po, c, nsf, ts_bytes):
    """packets of `ts_bytes` that are transmitted packets (any position) / all"""
    sent = po.stream_ts(c, 0, nsf, SEED_TS).reshape(-1, 188)
    idx = {bytes(p) for p in sent}
    got = np.asarray(ts_bytes).reshape(-1, 188)
    good = sum(1 for p in got if bytes(p) in idx)
    return len(got), good


def pre_viterbi_ber(c, noisy, clean, k=6):
    """bit errors of the first symbols that left the demodulator against the noise-free run's: every noisy symbol is matched with the clean symbol
    it agrees best with (the two runs may start at different superframes, and a lock period may be a few symbols long); median over k symbols"""
    if len(noisy) == 0 or len(clean) == 0:
        return None
    pop = np.array([bin(i).count("1") for i in range(256)], np.uint8)
    out = []
    for s in range(min(k, len(noisy))):
        e = pop[np.bitwise_xor(clean, noisy[s][None, :])].sum(axis=1)
        out.append(float(e.min()) / (clean.shape[1] * c.m))
    return float(np.median(out))


def sweep(po, g, nsf=3, snrs=SNRS, time_rs=True):
    import torch
    const, cr, mode = g.QPSK, g.C7_8, g.T8k
    c = po.cfg(const, cr, mode)
    clean = po.stream_slice(c, nsf, SEED_TS)
    oc = po.rx(c, clean, want=("bitdeint",))
    out = []
    for snr in snrs:
        iq = po.channel(clean, c.N, snr_db=snr, seed=SEED_NOISE)
        o = po.rx(c, iq, snr_db=snr, want=("bitdeint", "vit", "rs", "ts"))
        rx = g.Rx(const, cr, mode, max_samples=len(iq), snr_db=snr)
        rep = rx.run(iq)
        rs, ts, vit = rx.tap(g.TAP_RS), rx.tap(g.TAP_TS), rx.tap(g.TAP_VITERBI)
        row = {"snr_db": snr, "superframes": nsf, "samples": int(len(iq)),
               "oracle": {"symbols": int(o["n_acquired"]), "lock_periods": int(o["truncated"] + 1 if o["first_out_symbol"] >= 0 else 0),
                          "rs_bytes": int(len(o["rs"])), "ts_bytes": int(len(o["ts"])), "rs_fail_words": int(o["rs_fail"]), "rs_corrected_symbols": int(o["rs_corr"])},
               "hip": {"symbols": int(rep.total_symbols), "lock_periods": int(rep.n_lock_periods), "rs_bytes": int(len(rs)), "ts_bytes": int(len(ts)),
                       "rs_fail_words": int(rep.rs_fail_words), "rs_corrected_symbols": int(rep.rs_corrected_symbols)}}
        row["pre_viterbi_ber"] = pre_viterbi_ber(c, o["bitdeint"], oc["bitdeint"]) if o["first_out_symbol"] >= 0 else None
        for side, tsb in (("oracle", o["ts"]), ("hip", ts)):
            n, good = packet_stats(po, c, nsf, tsb)
            row[side]["ts_packets"], row[side]["good_packets"] = n, good
            row[side]["packet_error_rate"] = None if n == 0 else round(1.0 - good / n, 6)
        same_len = len(vit) == len(o["vit"]) and len(rs) == len(o["rs"]) and len(ts) == len(o["ts"])
        row["same_lengths"] = bool(same_len)
        row["viterbi_bytes_differing"] = int((vit != o["vit"]).sum()) if same_len else None
        row["rs_bytes_differing"] = int((rs != o["rs"]).sum()) if same_len else None
        row["ts_bytes_differing"] = int((ts != o["ts"]).sum()) if same_len else None
        if time_rs:
            # the whole synchronous run (every lock period, host round trips between them included) on resident samples
            dev = torch.from_numpy(iq.view(np.float32)).cuda()
            torch.cuda.synchronize()
            import time
            rx.run_device(dev.data_ptr(), len(iq))
            t0 = time.perf_counter()
            for _ in range(3):
                rx.run_device(dev.data_ptr(), len(iq))
            dt = (time.perf_counter() - t0) / 3
            row["hip"]["run_device_ms"] = round(dt * 1e3, 3)
            row["hip"]["msamples_per_s"] = round(len(iq) / dt / 1e6, 1)
        rx.close()
        out.append(row)
    return out


if __name__ == "__main__":
    import torch  # noqa: F401  (torch's HIP runtime first)
    from oracle import pyoracle as po
    import gr_dvbt_amd as g
    for row in sweep(po, g, int(sys.argv[1]) if len(sys.argv) > 1 else 3):
        print(json.dumps(row), flush=True)
