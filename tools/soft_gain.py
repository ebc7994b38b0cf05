"""Soft-decision mode (dvbt_rx_params.soft_decision) against the hard path on identical noisy samples: RS statistics and packet error rate per SNR.
`python tools/soft_gain.py` on the GPU box prints one JSON line per (configuration, channel, SNR)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def one(po, g, const, cr, mode, nsf, snr, echoes=(), seed=9, timing=False, rx_snr=None):
    c = po.cfg(const, cr, mode)
    clean = po.stream_slice(c, nsf, seed)
    iq = po.channel(clean, c.N, echoes=echoes, snr_db=snr, seed=5) if snr is not None else (po.channel(clean, c.N, echoes=echoes) if echoes else clean)
    sent = {bytes(p) for p in po.stream_ts(c, 0, nsf, seed).reshape(-1, 188)}
    row = {"constellation": int(const), "code_rate": int(cr), "mode": int(mode), "snr_db": snr, "echoes": [[d, [float(np.real(a)), float(np.imag(a))]] for d, a in echoes]}
    taps = {}
    for soft in (0, 1):
        rx = g.Rx(const, cr, mode, max_samples=len(iq), snr_db=rx_snr if rx_snr is not None else 30.0 if snr is None else snr, soft_decision=soft)
        rep = rx.run(iq)
        ts = rx.tap(g.TAP_TS).copy()
        pk = ts.reshape(-1, 188)
        good = sum(1 for p in pk if bytes(p) in sent)
        row["soft" if soft else "hard"] = {"lock_periods": int(rep.n_lock_periods), "rs_fail_words": int(rep.rs_fail_words), "rs_corrected_symbols": int(rep.rs_corrected_symbols),
                                           "ts_packets": int(len(pk)), "packet_error_rate": None if len(pk) == 0 else round(1 - good / len(pk), 5),
                                           "samples": int(len(iq))}
        if timing:
            import torch, time
            dev = torch.from_numpy(iq.view(np.float32)).cuda(); torch.cuda.synchronize()
            rx.enable_timing(True)
            for _ in range(3):
                rx.enqueue_device(dev.data_ptr(), len(iq)); rx.finish()
            row["soft" if soft else "hard"].update({"viterbi_stage_ms": round(rx.stage_ms("viterbi"), 3), "inner_stage_ms": round(rx.stage_ms("inner"), 3),
                                                    "fft_stage_ms": round(rx.stage_ms("fft"), 3), "total_ms": round(rx.stage_ms("total"), 3),
                                                    "msamples_per_s": round(len(iq) / rx.stage_ms("total") / 1e3, 1)})
        taps[soft] = ts
        rx.close()
    row["ts_identical"] = bool(len(taps[0]) == len(taps[1]) and (taps[0] == taps[1]).all())
    return row


if __name__ == "__main__":
    import torch  # noqa: F401
    from oracle import pyoracle as po
    import gr_dvbt_amd as g
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "awgn"):
        # the waterfall of the two decoders, half a dB apart
        for const, cr, mode, nsf, snrs in ((g.QAM16, g.C1_2, g.T2k, 8, [13.0 - 0.5 * i for i in range(13)]),
                                           (g.QAM64, g.C7_8, g.T8k, 3, [24.0 - 0.5 * i for i in range(12)])):
            for snr in snrs:
                print(json.dumps(one(po, g, const, cr, mode, nsf, snr)), flush=True)
    if which in ("all", "echo"):
        # a frequency-selective channel (echo of -6 dB at 20 samples: notches 9.5 dB below the peaks every 102 carriers), where the channel-state weights
        # matter.  ofdm_sym_acquisition's snr parameter at 10 dB: with the echo inside the guard interval the CP metric's peak is lower and wider, and the
        # peak detector of the reference drops the lock within a few hundred symbols at the demo flowgraphs' 30 dB
        for snr in [22.0 - 1.0 * i for i in range(13)]:
            print(json.dumps(one(po, g, g.QAM16, g.C3_4, g.T2k, 8, snr, echoes=((20, 0.5 + 0j),), rx_snr=10.0)), flush=True)
    if which in ("all", "timing"):
        print(json.dumps(one(po, g, g.QAM64, g.C7_8, g.T8k, 17, None, timing=True)), flush=True)      # throughput of the soft decoder on a clean 17-superframe segment
        print(json.dumps(one(po, g, g.QAM16, g.C1_2, g.T2k, 33, None, timing=True)), flush=True)
