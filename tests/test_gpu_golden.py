"""The HIP path (through the C ABI) against the committed golden vectors of tests/golden/.

viterbi_ref.npz = outputs of the reference's own d_viterbi.c kernels; chain_taps.json / chain_slices.npz =
taps of the pinned oracle on seeded loopbacks.  Nothing here reads /root/reference."""
import json
import os
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, G)
import make_golden as mg  # noqa: E402

TAPS = json.load(open(os.path.join(G, "chain_taps.json")))
CASES = {c[0]: c for c in mg.CHAIN_CASES}


@pytest.fixture(scope="module")
def g():
    import gr_dvbt_amd
    assert gr_dvbt_amd.device_count() > 0
    return gr_dvbt_amd


@pytest.mark.parametrize("const,cr", [(c[0], c[1]) for c in mg.VIT_CASES])
def test_viterbi_block_equals_reference_output(g, const, cr):
    v = np.load(os.path.join(G, "viterbi_ref.npz"))
    packed, want = v["c%d_r%d_in" % (const, cr)], v["c%d_r%d_out" % (const, cr)]
    d = g.get_dims(const, cr, 0)
    d_nsym, d_nout = 768 * d.cr_n // d.m, 768 * d.cr_k // 8
    nb = len(packed) // d_nsym
    b = g.Block("viterbi_decoder", const, 0, cr, 768, 0, -1)
    out = np.zeros(nb * d_nout, np.uint8)
    r, cons, _ = b.work(nb * d_nout, nb * d_nsym, packed[:nb * d_nsym].copy(), out,
                        tags=[(0, g.TAG_SUPERFRAME_START, 0xaa)])
    assert r == len(want) == nb * d_nout - d.ntraceback and cons == nb * d_nsym
    assert (out[:r] == want).all()
    b.close()


@pytest.mark.parametrize("name", sorted(CASES))
def test_chain_taps_equal_golden(po, g, name):
    _, const, cr, mode, nsf, seed, lead, snr = CASES[name]
    e = TAPS[name]
    c, iq = mg.make_case(const, cr, mode, nsf, seed, lead, snr)
    assert mg.sha(iq) == e["iq_sha256"]
    rx = g.Rx(const, cr, mode, max_samples=len(iq), taps=True, snr_db=30.0 if snr is None else snr)
    rep = rx.run(iq)
    assert rep.n_symbols == e["n_acquired"] and rep.first_out_symbol == e["first_out_symbol"]
    assert mg.sha(rx.tap(g.TAP_CP_START).astype(np.int32)) == e["cp_start_sha256"]
    tapid = {"demap": g.TAP_DEMAP, "symdeint": g.TAP_SYMDEINT, "bitdeint": g.TAP_BITDEINT,
             "vit": g.TAP_VITERBI, "deint": g.TAP_DEINT, "rs": g.TAP_RS, "ts": g.TAP_TS}
    # Clean loopback: every integer tap bit-exact.  Under AWGN the hard decisions before the decoder sit in
    # the float-tolerance domain (a carrier on a decision boundary may flip); the decoded taps must still be
    # identical, which is the contract of BASELINE config 5 (post-RS comparison on the identical noisy file).
    names = mg.INT_TAPS if snr is None else ("rs", "ts")
    for t in names:
        a = rx.tap(tapid[t]).reshape(-1)
        assert a.size == e["taps"][t]["n"], t
        assert mg.sha(a) == e["taps"][t]["sha256"], t
    assert rep.rs_fail_words == e["rs_fail"]
    if snr is None:
        assert rep.rs_corrected_symbols == e["rs_corr"]
        s = np.load(os.path.join(G, "chain_slices.npz"))
        if name + "_eq" in s:
            f = e["first_out_symbol"]
            fft = rx.tap(g.TAP_FFT)[f:f + 1]
            assert np.abs(fft - s[name + "_fft"]).max() <= 1e-5 * np.abs(s[name + "_fft"]).max()
            eq = rx.tap(g.TAP_EQ)[0:2]
            dd = eq - s[name + "_eq"]
            assert max(np.abs(dd.real).max(), np.abs(dd.imag).max()) <= 1e-3 * 2 * c.norm
            assert (rx.tap(g.TAP_VITERBI)[:1344] == s[name + "_vit"]).all()
            assert (rx.tap(g.TAP_TS)[:1504] == s[name + "_ts"]).all()
    rx.close()


CH_CASES = {c[0]: c for c in mg.CHANNEL_CASES}


@pytest.mark.parametrize("name", sorted(CH_CASES))
def test_channel_taps_equal_golden(po, g, name):
    """carrier offsets, echoes, lock losses: the HIP path against the committed hashes (lock periods, per-symbol CP position / integer offset / symbol
    index, every integer tap; noisy case: the decoded taps)"""
    _, const, cr, mode, nsf, seed, chan = CH_CASES[name]
    e = TAPS[name]
    c, iq = mg.make_channel_case(const, cr, mode, nsf, seed, chan)
    assert mg.sha(iq) == e["iq_sha256"]
    rx = g.Rx(const, cr, mode, max_samples=len(iq), taps=True, snr_db=30.0 if e["snr_db"] is None else e["snr_db"])
    rep = rx.run(iq)
    L = c.N + c.cp
    assert rep.total_symbols == e["n_acquired"] and rep.n_lock_periods == e["lock_periods_delivering"]
    assert [[off + fc * L, n] for (off, fc, cp0, n, fo) in rx.lock_periods() if n > 0] == e["lock_periods"]
    single = len(e["lock_periods"]) == 1
    if single:
        ns = rep.n_symbols - 1
        assert rep.first_out_symbol == e["first_out_symbol"]
        assert mg.sha(rx.tap(g.TAP_CP_START).astype(np.int32)) == e["cp_start_sha256"]
    tapid = {"demap": g.TAP_DEMAP, "symdeint": g.TAP_SYMDEINT, "bitdeint": g.TAP_BITDEINT,
             "vit": g.TAP_VITERBI, "deint": g.TAP_DEINT, "rs": g.TAP_RS, "ts": g.TAP_TS}
    names = ("rs", "ts") if e["snr_db"] is not None else (mg.INT_TAPS if single else ("vit", "deint", "rs", "ts"))   # several periods: the per-symbol taps hold the last period only
    for t in names:
        a = rx.tap(tapid[t]).reshape(-1)
        assert a.size == e["taps"][t]["n"], t
        if a.size:
            assert mg.sha(a) == e["taps"][t]["sha256"], t
    assert rep.rs_fail_words == e["rs_fail"]
    rx.close()


@pytest.mark.parametrize("compat", [0, 1])
def test_rs_words_equal_golden(g, compat):
    s = np.load(os.path.join(G, "chain_slices.npz"))
    words = np.ascontiguousarray(s["rs_in"])
    b = g.Block("reed_solomon_dec", 2, 8, 0x11d, 255, 239, 8, 51, 8, compat)
    out = np.zeros((16, 188), np.uint8)
    r, cons, _ = b.work(2, 2, words, out)
    assert r == 2 and cons == 2
    assert (out == s["rs_out_compat%d" % compat]).all()
    b.close()


HIER = json.load(open(os.path.join(G, "hier_taps.json")))
HIER_CASES = {c[0]: c for c in mg.HIER_CASES}


@pytest.mark.parametrize("name", sorted(HIER_CASES))
def test_hierarchical_taps_equal_golden(po, g, name):
    """the chain in a hierarchical mode against the committed hashes: demapper decisions on the alpha grid, symbol de-interleaver, both outputs of the bit
    de-interleaver, the decoder's bytes on output 0"""
    _, const, hier, cr, mode, nsf, seed, lead = HIER_CASES[name]
    e = HIER[name]
    c, iq = mg.make_hier_case(const, hier, cr, mode, nsf, seed, lead)
    assert mg.sha(iq) == e["iq_sha256"]
    rx = g.Rx(const, cr, mode, max_samples=len(iq), hierarchy=hier, taps=True)
    rep = rx.run(iq)
    assert rep.n_symbols == e["n_acquired"] and rep.first_out_symbol == e["first_out_symbol"]
    tapid = {"demap": g.TAP_DEMAP, "symdeint": g.TAP_SYMDEINT, "bitdeint": g.TAP_BITDEINT, "bitdeint_lp": g.TAP_BITDEINT_LP, "vit": g.TAP_VITERBI}
    for t in mg.HIER_TAPS:
        a = rx.tap(tapid[t]).reshape(-1)
        assert a.size == e["taps"][t]["n"] and mg.sha(a) == e["taps"][t]["sha256"], t
    rx.close()
