"""The front end on what a real capture looks like: carrier offset (fraction of a subcarrier -> ofdm_sym_acquisition's epsilon and
derotation, ofdm_sym_acquisition_impl.cc:277-312; whole subcarriers -> pilot_gen::process_cpilot_data's search over [-8, +7] and the shifted
carrier read, reference_signals_impl.cc:715-744, 793-819) and a static multipath channel (the equaliser's interpolation between the
estimation carriers of a frequency-selective channel, reference_signals_impl.cc:536-689).  HIP path against the oracle on the same samples
(oracle/pyoracle.py::channel builds them in double precision):

* clean channels: lock decisions (cp_start of every symbol), integer offset and symbol index of every demodulated symbol identical, the
  equalised-carrier tap within 1e-3 * 2 * d_norm per component, every integer tap from the demapper on bit-exact;
* with AWGN on top: post-RS output and failure count identical;
* channels on which the reference's peak detector loses the CP lock again and again (an echo at -6 dB flattens the tracking window's lambda
  so that the fall below 0.9 d_avg is missed, ofdm_sym_acquisition_impl.cc:72-146): the same lock periods, the same bytes (possibly none);
* through the segment API and through the ten blocks driven call by call.

The ACQ / FFT taps are compared per symbol up to a common phasor (the accumulated deviation of the reference's float phase accumulator from the exact
line at the symbol's entry is one rotation per symbol, which the equaliser divides out); the wander INSIDE a symbol is reproduced (k_drift.hpp)."""
import numpy as np
import pytest

import gr_dvbt_amd as g
from gr_dvbt_amd.flowgraph import RxFlowgraph

pytestmark = pytest.mark.gpu

INT_TAPS = (("demap", g.TAP_DEMAP), ("symdeint", g.TAP_SYMDEINT), ("bitdeint", g.TAP_BITDEINT), ("vit", g.TAP_VITERBI),
            ("deint", g.TAP_DEINT), ("rs", g.TAP_RS), ("ts", g.TAP_TS))
WANT = ("acq", "fft", "eq") + tuple(k for k, _ in INT_TAPS)

# A1 / A2 taps: the reference accumulates the derotation phase in a FLOAT, one addition per sample (:285-309); inside a binade every step rounds the same
# way, so the phase runs at a slightly wrong rate that changes from binade to binade -- up to 5e-4 rad of wander inside one 8k symbol
# (tests/test_phase_accumulator_model.py).  The kernels reproduce that wander in closed form (k_drift.hpp, tests/test_drift_model.py) whenever the
# offset has a usable fractional part: the taps then agree to ~2e-6 of the peak and the EQ tap to ~1e-5 of the constellation spacing (2.4e-3 .. 3.7e-3
# without it: above the contract's 1e-3).  Where the model does not apply (estimates that jitter around zero: echoes without an offset, |epsilon| < 4e-3)
# the accumulator lives near zero, where its grid is fine; what is left is below 1.2e-4 of the peak, which this bound allows.
TOL_DEROT = 2e-4

K2 = (g.QAM16, g.C1_2, g.T2k, 4)         # BASELINE config 2
K8 = (g.QAM64, g.C7_8, g.T8k, 2)         # BASELINE config 3


def _stream(po, cfgt, seed=9, **chan):
    const, cr, mode, nsf = cfgt
    c = po.cfg(const, cr, mode)
    return c, po.channel(po.stream_slice(c, nsf, seed), c.N, **chan)


def _unrotate(a, b):
    """a, b: [symbols][N]; b rotated per symbol onto a (least squares), both as arrays"""
    ph = (a * np.conj(b)).sum(axis=1)
    ph = ph / np.maximum(np.abs(ph), 1e-30)
    return b * ph[:, None]


# cp = 64 @2k, 256 @8k: echoes at 0.3 cp and 0.9 cp; amplitudes at which the reference's tracker still holds (see the lock-loss test for -6 dB)
CLEAN = [
    ("2k cfo +0.37", K2, dict(cfo=0.37)),
    ("2k cfo -0.37", K2, dict(cfo=-0.37)),
    ("2k cfo +3", K2, dict(cfo=3.0)),
    ("2k cfo -5", K2, dict(cfo=-5.0)),
    ("2k cfo -8 + 0.37 (lower edge of the search)", K2, dict(cfo=-7.63)),
    ("2k cfo +7 + 0.2 (upper edge of the search)", K2, dict(cfo=7.2)),
    ("2k echo 0.3 cp -10 dB", K2, dict(echoes=((19, 0.3),))),
    ("2k echo 0.9 cp -14 dB, rotated", K2, dict(echoes=((57, 0.2j),))),
    ("2k two echoes + cfo 3.37", K2, dict(echoes=((57, 0.15), (19, -0.1j)), cfo=3.37)),
    ("8k cfo +0.37", K8, dict(cfo=0.37)),
    ("8k cfo -5 - 0.37", K8, dict(cfo=-5.37)),
    ("8k cfo -7.63", K8, dict(cfo=-7.63)),
    ("8k cfo +7.2", K8, dict(cfo=7.2)),
    ("8k cfo +0.01", K8, dict(cfo=0.01)),                 # a small constant offset: the float accumulator crawls through its coarse binades
    ("8k cfo -0.0007", K8, dict(cfo=-0.0007)),            # at the edge of what k_drift.hpp models
    ("8k cfo +0.0001", K8, dict(cfo=0.0001)),             # below it: granular steps of the accumulator, nothing applied
    ("8k echo 0.3 cp -16 dB", K8, dict(echoes=((77, 0.15),))),
    ("8k echo 0.3 cp -20 dB + cfo 2.2", K8, dict(echoes=((77, 0.1),), cfo=2.2)),
]


@pytest.mark.parametrize("name,cfgt,chan", CLEAN, ids=[c[0] for c in CLEAN])
def test_clean_channel_every_tap(po, name, cfgt, chan):
    c, iq = _stream(po, cfgt, **chan)
    o = po.rx(c, iq, want=WANT)
    assert o["truncated"] == 0 and o["first_out_symbol"] >= 0 and len(o["ts"]) > 0, "the case is meant to hold the lock"
    rx = g.Rx(cfgt[0], cfgt[1], cfgt[2], max_samples=len(iq), taps=True)
    rep = rx.run(iq)
    assert rep.n_lock_periods == 1 and rep.n_symbols == o["n_acquired"] and rep.first_out_symbol == o["first_out_symbol"]
    assert (rx.tap(g.TAP_CP_START) == o["cp_start"]).all()
    ns = rep.n_symbols - 1
    assert (rx.tap(g.TAP_SYMBOL_INDEX) == o["sym_index"][:ns]).all()
    fo = rx.tap(g.TAP_FREQ_OFFSET)
    assert (fo == o["freq_offset"][:ns]).all()
    want_fo = int(np.round(chan.get("cfo", 0.0)))
    assert (fo[rep.first_out_symbol:ns - 4] == want_fo).all()        # and it is the offset that was applied
    # float taps: A1 / A2 up to one rotation per symbol, A3 within the stated tolerance
    acq, fft, eq = rx.tap(g.TAP_ACQ), rx.tap(g.TAP_FFT), rx.tap(g.TAP_EQ)
    assert acq.shape == o["acq"].shape and fft.shape == o["fft"].shape and eq.shape == o["eq"].shape
    e_acq = np.abs(_unrotate(acq, o["acq"]) - acq).max() / np.abs(acq).max()
    e_fft = np.abs(_unrotate(fft, o["fft"]) - fft).max() / np.abs(fft).max()
    d = eq - o["eq"]
    e_eq = max(np.abs(d.real).max(), np.abs(d.imag).max()) / (2 * c.norm)
    print(f"\n[{name}] acq {e_acq:.2e} fft {e_fft:.2e} of the peak; eq {e_eq:.2e} of the constellation spacing (tolerance 1e-3)")
    assert e_acq <= TOL_DEROT and e_fft <= TOL_DEROT
    assert e_eq <= 1e-3
    for key, tap in INT_TAPS:
        a, b = rx.tap(tap).reshape(-1), o[key].reshape(-1)
        assert a.size == b.size > 0, key
        assert (a == b).all(), key
    assert rep.rs_fail_words == o["rs_fail"] and rep.rs_corrected_symbols == o["rs_corr"]
    rx.close()


def test_offset_outside_the_search_range(po):
    """+8 subcarriers: process_cpilot_data searches [-8, +7] (:724), so the continual pilots are never found and nothing is decoded -- by both"""
    c, iq = _stream(po, K2, cfo=8.2)
    o = po.rx(c, iq, want=("ts",))
    rx = g.Rx(K2[0], K2[1], K2[2], max_samples=len(iq), taps=True)
    rep = rx.run(iq)
    assert o["first_out_symbol"] == -1 and rep.first_out_symbol == -1 and rep.n_ts_bytes == 0 and rep.status & 4
    assert rep.total_symbols == o["n_acquired"]
    if rep.n_symbols == o["n_acquired"]:
        assert (rx.tap(g.TAP_FREQ_OFFSET) == o["freq_offset"][:rep.n_symbols - 1]).all()
    rx.close()


NOISY = [
    ("2k echo 0.9 cp -14 dB + 25 dB AWGN", K2, dict(echoes=((57, 0.2),), snr_db=25.0)),
    ("2k echo 0.3 cp -10 dB + cfo -2.37 + 25 dB", K2, dict(echoes=((19, 0.3),), cfo=-2.37, snr_db=25.0)),
    ("8k echo 0.3 cp -16 dB + 28 dB AWGN", K8, dict(echoes=((77, 0.15),), snr_db=28.0)),
    ("8k cfo +3.37 + 25 dB AWGN", K8, dict(cfo=3.37, snr_db=25.0)),
]


@pytest.mark.parametrize("name,cfgt,chan", NOISY, ids=[c[0] for c in NOISY])
def test_noisy_channel_post_rs_identical(po, name, cfgt, chan):
    c, iq = _stream(po, cfgt, **chan)
    snr = chan["snr_db"]
    o = po.rx(c, iq, snr_db=snr, want=("bitdeint", "vit", "rs", "ts"))
    assert len(o["rs"]) > 0
    rx = g.Rx(cfgt[0], cfgt[1], cfgt[2], max_samples=len(iq), snr_db=snr, taps=True)
    rep = rx.run(iq)
    assert rep.total_symbols == o["n_acquired"] and rep.n_lock_periods == o["truncated"] + 1
    if o["truncated"] == 0:
        assert (rx.tap(g.TAP_CP_START) == o["cp_start"]).all()
        assert (rx.tap(g.TAP_FREQ_OFFSET) == o["freq_offset"][:rep.n_symbols - 1]).all()
        bd = rx.tap(g.TAP_BITDEINT)
        assert bd.shape == o["bitdeint"].shape and (bd != o["bitdeint"]).mean() < 2e-5   # a noisy point within float rounding of a decision boundary
    for key, tap in (("rs", g.TAP_RS), ("ts", g.TAP_TS)):
        a = rx.tap(tap)
        assert a.size == o[key].size, key
        assert (a == o[key]).all(), key
    assert rep.rs_fail_words == o["rs_fail"]
    rx.close()


LOSSY = [
    ("2k echo 0.9 cp -6 dB", (g.QAM16, g.C1_2, g.T2k, 6), dict(echoes=((57, 0.5),))),
    ("2k echo 0.3 cp -6 dB", (g.QAM16, g.C1_2, g.T2k, 6), dict(echoes=((19, 0.5),))),
    ("2k echo 0.3 cp -8 dB", (g.QAM16, g.C1_2, g.T2k, 6), dict(echoes=((19, 0.4),))),
    ("2k two echoes -14 / -20 dB + cfo 3.37", K2, dict(echoes=((57, 0.2), (19, -0.1j)), cfo=3.37)),
    ("2k echoes + cfo + 22 dB", (g.QAM16, g.C1_2, g.T2k, 4), dict(echoes=((57, 0.2), (19, -0.1j)), cfo=3.37, snr_db=22.0)),
    ("8k echo 0.9 cp -20 dB", K8, dict(echoes=((230, 0.1),))),
    ("8k echo 0.3 cp -14 dB", K8, dict(echoes=((77, 0.2),))),
    ("8k echo 0.9 cp -6 dB + 25 dB", K8, dict(echoes=((230, 0.5),), snr_db=25.0)),
    ("8k echo rotated + cfo -2.37", K8, dict(echoes=((77, 0.2j),), cfo=-2.37)),
]


@pytest.mark.parametrize("name,cfgt,chan", LOSSY, ids=[c[0] for c in LOSSY])
def test_channels_that_break_the_cp_lock(po, name, cfgt, chan):
    """what the reference does on these inputs is lose and regain the CP lock; the library follows it through every period"""
    c, iq = _stream(po, cfgt, **chan)
    snr = chan.get("snr_db") or 30.0
    o = po.rx(c, iq, snr_db=snr, want=("vit", "rs", "ts"))
    rx = g.Rx(cfgt[0], cfgt[1], cfgt[2], max_samples=len(iq), snr_db=snr)
    rep = rx.run(iq)
    assert rep.total_symbols == o["n_acquired"], (rep.total_symbols, o["n_acquired"])
    # every lock period: the sample at which the call that delivered its first item began, and how many items it delivered
    L = c.N + c.cp
    assert [(off + fc * L, n) for (off, fc, cp0, n, fo) in rx.lock_periods() if n > 0] == o["lock_periods"]
    delivered = o["truncated"] + 1 if o["first_out_symbol"] >= 0 else 0
    assert rep.n_lock_periods == delivered
    assert rep.n_viterbi_bytes == len(o["vit"]) and rep.n_rs_bytes == len(o["rs"]) and rep.n_ts_bytes == len(o["ts"])
    if delivered:
        for key, tap in (("vit", g.TAP_VITERBI), ("rs", g.TAP_RS), ("ts", g.TAP_TS)):
            if "snr_db" in chan and key == "vit":
                continue
            a = rx.tap(tap)
            assert a.size == o[key].size and (a == o[key]).all(), key
        assert rep.rs_fail_words == o["rs_fail"]
    rx.close()


FLOW = [
    ("2k cfo -5.37", K2, dict(cfo=-5.37), "device", 4),
    ("2k echo + cfo", K2, dict(echoes=((19, 0.3),), cfo=3.0), "host", 4),
    ("8k cfo +7.2", K8, dict(cfo=7.2), "device", 33),
    ("8k echo 0.3 cp -16 dB", K8, dict(echoes=((77, 0.15),)), "device", 4),
    ("8k cfo -0.37", K8, dict(cfo=-0.37), "device", 16),
    ("8k cfo +0.01", K8, dict(cfo=0.01), "device", 8),
]


@pytest.mark.parametrize("name,cfgt,chan,mode,call_symbols", FLOW, ids=[c[0] for c in FLOW])
def test_block_by_block_on_a_real_channel(po, name, cfgt, chan, mode, call_symbols):
    """the drop-in path: ofdm_sym_acquisition -> fft -> demod_reference_signals -> ... driven call by call (k_frontend.hpp::demod_kernel
    applies frequency_correction's phasor and reads the next item, unlike the fused symbol kernels)"""
    c, iq = _stream(po, cfgt, **chan)
    o = po.rx(c, iq, want=("acq", "eq", "ts"))
    ref = o["ts"]
    fg = RxFlowgraph(cfgt[0], cfgt[1], cfgt[2], len(iq), mode=mode, call_symbols=call_symbols)
    ts = fg.run(iq)

    def items(k, dtype, width):
        st = fg.stages[k]
        buf = st.out[:st.produced * st.out_item]
        buf = buf.cpu().numpy() if mode == "device" else np.asarray(buf)
        return buf.view(dtype).reshape(-1, width)
    acq, eq = items(0, np.complex64, c.N), items(2, np.complex64, c.payload)
    fg.close()
    n = min(len(ts), len(ref))
    assert n > 0.9 * len(ref) and abs(len(ts) - len(ref)) <= 64 * 1504
    assert (ts[:n] == ref[:n]).all()
    # the float taps of the single blocks: ofdm_sym_acquisition's items up to one rotation per item, demod_reference_signals' items within the
    # contract's tolerance (the tracker of the block carries the float phase accumulator's own value from call to call: k_drift.hpp)
    na = min(len(acq), len(o["acq"]))
    assert na > 0.9 * len(o["acq"])
    e_acq = np.abs(_unrotate(acq[:na], o["acq"][:na]) - acq[:na]).max() / np.abs(acq[:na]).max()
    ne = min(len(eq), len(o["eq"]))
    assert ne > 0.9 * len(o["eq"])
    d = eq[:ne] - o["eq"][:ne]
    e_eq = max(np.abs(d.real).max(), np.abs(d.imag).max()) / (2 * c.norm)
    print(f"\n[blocks: {name}] acq {e_acq:.2e} of the peak; eq {e_eq:.2e} of the constellation spacing (tolerance 1e-3)")
    assert e_acq <= TOL_DEROT and e_eq <= 1e-3
