"""Throughput of the streaming entry (dvbt_rx_stream_push / pull) at GNU Radio-like call sizes: host samples pushed `symbols` OFDM symbols per call."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def run(po, g, workload=("QAM64", "C7_8", "T8k"), nsf=33, seg_sf=8, symbols=64, device=False, reps=2, awgn_db=None, verify=False, borrow=0):
    """awgn_db: AWGN at that SNR, ofdm_sym_acquisition's snr set to it (BASELINE config 5 at the prescribed noise: the reference's tracker drops the lock every few
    dozen symbols and the stream is walked window by window); verify: the TS pulled is compared with dvbt_rx_segment_run over the whole stream"""
    import torch
    const, cr, mode = (getattr(g, x) for x in workload)
    c = po.cfg(const, cr, mode)
    iq = po.stream_slice(c, nsf, 77)
    snr = 30.0
    if awgn_db is not None:
        iq = po.channel(iq, c.N, snr_db=awgn_db, seed=5); snr = float(awgn_db)
    L = c.N + c.cp
    step = symbols * L
    dev = torch.from_numpy(iq.view(np.float32)).cuda() if device else None
    torch.cuda.synchronize()
    best, nbytes = None, 0
    for _ in range(reps):
        st = g.RxStream(const, cr, mode, segment_superframes=seg_sf, snr_db=snr, borrow=borrow)
        t0 = time.perf_counter()
        got = 0; parts = []
        for a in range(0, len(iq), step):
            n = min(step, len(iq) - a)
            if device:
                st.push_device(dev.data_ptr() + 8 * a, n)
            else:
                st.push(iq[a:a + n])
            parts.append(st.pull()); got += len(parts[-1])
        st.finish()
        parts.append(st.pull()); got += len(parts[-1])
        dt = time.perf_counter() - t0
        info = st.info()
        st.close()
        best = dt if best is None or dt < best else best
        nbytes = got
    row = {"entry": ("dvbt_rx_stream_push_device, samples borrowed (borrow_device_pushes)" if borrow else "dvbt_rx_stream_push_device") if device else "dvbt_rx_stream_push (host samples)", "symbols_per_call": symbols, "segment_superframes": seg_sf,
           "stream_superframes": nsf, "samples": int(len(iq)), "seconds": round(best, 4), "value": round(len(iq) / best / 1e6, 1), "unit": "Msamples/s",
           "x_realtime": round(len(iq) / best / 1e6 / (64 / 7), 1), "ts_bytes": int(nbytes), "status": int(info.status)}
    if awgn_db is not None:
        row["workload"] = "%s %s %s + AWGN %g dB (snr parameter = channel)" % (workload + (awgn_db,)); row["in_walk_at_the_end"] = int(info.in_walk)
    if verify:
        rx = g.Rx(const, cr, mode, max_samples=len(iq), snr_db=snr)
        rx.run(iq)
        ref = rx.tap(g.TAP_TS)
        ts = np.concatenate(parts)
        row["equals_the_single_chain"] = bool(len(ts) == len(ref) and (ts == ref).all()); row["lock_periods_that_delivered"] = int(rx.report.n_lock_periods)
        rx.close()
    return row


if __name__ == "__main__":
    import torch  # noqa: F401
    from oracle import pyoracle as po
    import gr_dvbt_amd as g
    for sym, dev in ((4, False), (64, False), (64, True), (256, False)):
        print(json.dumps(run(po, g, symbols=sym, device=dev)), flush=True)
