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


# per guard interval: the noise level at which the reference's peak detector loses and finds the lock again and again (scanned with the oracle: QPSK 1/2, seed 7);
# 2k with a guard interval of 1/8 holds its lock at every noise level down to 3 dB: there the losses come from three dropouts of 30 symbols
WALK_CASES = [(0, 1, 4.0, (), 16), (0, 2, 12.0, (300, 700, 1100), 16), (0, 3, 8.0, (), 10), (1, 1, 11.0, (), 10), (1, 2, 12.0, (), 4), (1, 3, 12.0, (), 2), (0, 0, 8.0, (), 16)]


@pytest.mark.parametrize("mode,guard,snr,holes,chunk", WALK_CASES, ids=["2k 1/16", "2k 1/8", "2k 1/4", "8k 1/16", "8k 1/8", "8k 1/4", "2k 1/32"])
def test_lock_period_walk_on_every_guard_interval(po, mode, guard, snr, holes, chunk):
    """The lock-period walk's one-launch tracker (acq_small_kernel: chunks of 16 calls for a short guard interval down to 2 for cp = 2048, the products of a call's
    lags staged in LDS) on a stream whose CP lock is lost again and again, for every guard interval and both modes: the same lock periods (where each began, how many
    items it delivered) as the oracle's sequential restatement of ofdm_sym_acquisition, the same byte counts, and -- QPSK 1/2, where the RS decoder still corrects --
    the same bytes behind the RS decoder.  No case may skip: every guard interval has its channel (noise level, or dropouts where noise alone does not shake the
    lock), and the chunk size the one-launch tracker ran with is checked."""
    import gr_dvbt_amd as g
    import numpy as np
    c = po.cfg(po.QPSK, po.C1_2, mode, guard=guard)
    L = c.N + c.cp
    nsf = 3 if mode == 1 else 6
    clean = po.stream_slice(c, nsf, 31 + guard)
    iq = po.channel(clean, c.N, snr_db=snr, seed=7).copy()
    for hs in holes:
        a = po.STREAM_LEAD_IN + hs * L
        iq[a:a + 30 * L] = 0
    o = po.rx(c, iq, snr_db=snr, want=("vit", "rs", "ts"))
    assert len(o["lock_periods"]) >= 2, "the channel of this case is meant to shake the lock"
    rx = g.Rx(po.QPSK, po.C1_2, mode, max_samples=len(iq), guard=guard, snr_db=snr)
    rep = rx.run(iq)
    got = [(off + fc * L, n) for (off, fc, cp0, n, fo) in rx.lock_periods() if n > 0]
    assert got == o["lock_periods"], (len(got), len(o["lock_periods"]))
    small, general, chunk_calls, small_max = rx.walk_stats()
    assert chunk_calls == chunk, (chunk_calls, chunk)                 # what fits the LDS next to the lags' products for this cp
    if len(got) >= 4:
        assert small >= 1, (small, general)                         # the short look-ahead windows of a stream that keeps losing the lock go through the one-launch tracker
    assert rep.total_symbols == o["n_acquired"]
    assert rep.n_viterbi_bytes == len(o["vit"]) and rep.n_rs_bytes == len(o["rs"]) and rep.n_ts_bytes == len(o["ts"])
    if len(o["rs"]):
        a = rx.tap(g.TAP_RS)
        assert (a != o["rs"]).mean() < 1e-3                        # identical but for what failed RS words pass through (a decision within float rounding of a boundary)
    rx.close()


@pytest.mark.parametrize("snr", [9.0, 8.0])
def test_viterbi_stage_is_the_streaming_decoder_on_every_lock_period(po, snr):
    """BASELINE config 5 (8k QPSK 7/8 + AWGN) at the prescribed noise, where the Viterbi tap differs from the oracle's in a handful of bytes: is that the decoder, or
    its INPUT (a hard decision within float rounding of a boundary)?  The handle keeps every lock period's decoder input (dvbt_rx_enable_taps(h, 2)); per
    period the HIP decoder's bytes must equal the streaming decoder (oracle/o_viterbi.c, pinned to the reference's lib/d_viterbi.c) over THAT input -- all of them, with
    default parameters.  Whatever differs from the oracle's chain at the tap is then the input's few bits, counted here too."""
    import ctypes as C
    import numpy as np
    import gr_dvbt_amd as g
    import snr_sweep
    c = po.cfg(po.QPSK, po.C7_8, po.T8k)
    iq = po.channel(po.stream_slice(c, 3, snr_sweep.SEED_TS), c.N, snr_db=snr, seed=snr_sweep.SEED_NOISE)
    o = po.rx(c, iq, snr_db=snr, want=("bitdeint", "vit"))
    rx = g.Rx(po.QPSK, po.C7_8, po.T8k, max_samples=len(iq), snr_db=snr, taps=2, viterbi_verify=1)
    rep = rx.run(iq)
    vit = rx.tap(g.TAP_VITERBI)
    periods = rx.period_taps()
    assert len(periods) == rep.n_lock_periods >= 1 and rep.n_viterbi_bytes == len(o["vit"])
    po.lib().o_viterbi_decode.restype = C.c_size_t
    total = 0
    for i, (bd, voff, vbytes) in enumerate(periods):
        bd = np.ascontiguousarray(bd)
        ref = np.zeros(bd.size * c.m * c.k // (8 * c.n) + 64, np.uint8)
        n = po.lib().o_viterbi_decode(C.byref(c), 768, bd.ctypes.data_as(C.c_void_p), C.c_size_t(bd.size), ref.ctypes.data_as(C.c_void_p))
        assert n == vbytes, (n, vbytes)
        # (the next period's bytes start at a multiple of two de-interleaver items, 3264 bytes, at or below this period's end -- the reference's byte de-interleaver realigns
        # its input at the tag, convolutional_deinterleaver_impl.cc:109-120 -- and overwrite what lies behind)
        n = min(n, periods[i + 1][1] - voff) if i + 1 < len(periods) else n
        assert (vit[voff:voff + n] == ref[:n]).all(), ("lock period at Viterbi offset", voff, int((vit[voff:voff + n] != ref[:n]).sum()))
        total += n
    assert total > 100000
    # the decoder's input against the oracle chain's: the same symbols, a few bits apart (a demapper decision within float rounding of a boundary)
    mine = np.concatenate([bd for bd, _, _ in periods])
    theirs = o["bitdeint"].reshape(-1)
    assert mine.size == theirs.size
    nbits = int(np.unpackbits(mine ^ theirs).sum())
    print(f"config 5 at {snr} dB: {len(periods)} lock periods, {total} Viterbi bytes == the streaming decoder over the chain's own input; decoder-input bits that differ from the oracle chain's: "
          f"{nbits} of {mine.size * c.m}; Viterbi bytes that differ from the oracle chain's: {int((vit != o['vit']).sum())}; proof of the last launch: {rx.viterbi_proof()}")
    assert nbits <= 1e-4 * mine.size * c.m
    rx.close()
