"""BASELINE config 5 (8k QPSK 7/8 + AWGN) where it means something: an SNR sweep from 14 dB (error free behind the RS decoder) through 8 dB
(pre-Viterbi BER ~ 1e-2, the noise level SURVEY 8d prescribes: the RS decoder fails on most words, the CP lock drops six times in three
superframes) to 5 dB (nothing locks), HIP path vs oracle on the IDENTICAL noisy samples (tools/snr_sweep.py).  Per point: the same lock structure
(symbols acquired, lock periods that delivered, byte counts at every stream tap), the same packet error rate against the transmitted packets,
the same RS statistics; the bytes themselves identical wherever the RS decoder still corrects (>= 9 dB), and differing in a handful of bytes
where it has given up (a hard decision within float rounding of a boundary flips one Viterbi input bit in ~1e-5; failed RS words pass such
differences through, as the reference's do)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_snr_sweep_equals_the_oracle(po):
    import gr_dvbt_amd as g
    import snr_sweep
    rows = snr_sweep.sweep(po, g, nsf=3, snrs=(14.0, 10.0, 9.0, 8.0, 7.0, 5.0), time_rs=False)
    by = {r["snr_db"]: r for r in rows}
    for r in rows:
        o, h = r["oracle"], r["hip"]
        for k in ("symbols", "lock_periods", "rs_bytes", "ts_bytes", "ts_packets"):
            assert o[k] == h[k], (r["snr_db"], k, o[k], h[k])
        assert r["same_lengths"]
        assert o["rs_fail_words"] == h["rs_fail_words"] or abs(o["rs_fail_words"] - h["rs_fail_words"]) <= 2, r["snr_db"]
        if o["packet_error_rate"] is not None:
            assert abs(o["packet_error_rate"] - h["packet_error_rate"]) <= 2e-3, r["snr_db"]
        if o["rs_fail_words"] <= 11 * max(o["lock_periods"], 1):     # only the junction words of the lock periods fail: everything else is corrected
            assert r["rs_bytes_differing"] == 0 and r["ts_bytes_differing"] == 0, r["snr_db"]
        elif o["rs_bytes"]:
            assert r["rs_bytes_differing"] <= 1e-3 * o["rs_bytes"], r["snr_db"]
    # the sweep really crosses the waterfall: clean at 14 dB, the prescribed pre-Viterbi BER at 8 dB with the RS decoder overwhelmed, nothing at 5 dB
    assert by[14.0]["oracle"]["rs_corrected_symbols"] == 0 and by[14.0]["oracle"]["packet_error_rate"] < 0.01
    assert by[9.0]["oracle"]["rs_corrected_symbols"] > 1000 and by[9.0]["oracle"]["packet_error_rate"] < 0.05
    assert 5e-3 < by[8.0]["pre_viterbi_ber"] < 2e-2 and by[8.0]["oracle"]["packet_error_rate"] > 0.5
    assert by[8.0]["oracle"]["lock_periods"] >= 2
    assert by[5.0]["oracle"]["ts_bytes"] == 0
