"""The oracle against the committed golden vectors (tests/golden/, made by tests/golden/make_golden.py).

viterbi_ref.npz holds outputs of the REFERENCE's own d_viterbi.c kernels (oracle/_ref), so this test pins
the oracle's Viterbi block to the reference even where oracle/_ref is not present (the GPU box).  The chain
fixtures pin the oracle against drift: any change to oracle/ that alters a tap shows up here."""
import ctypes as C
import json
import os
import sys
import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, G)
import make_golden as mg  # noqa: E402  (the seeded input generator shared with the fixture script)
TAPS = json.load(open(os.path.join(G, "chain_taps.json")))
CASES = {c[0]: c for c in mg.CHAIN_CASES}


@pytest.mark.parametrize("const,cr", [(c[0], c[1]) for c in mg.VIT_CASES])
def test_oracle_viterbi_block_equals_reference_output(po, const, cr):
    v = np.load(os.path.join(G, "viterbi_ref.npz"))
    packed, want = v["c%d_r%d_in" % (const, cr)], v["c%d_r%d_out" % (const, cr)]
    c = po.cfg(const, cr, po.T2k)
    out = np.zeros(len(want) + 4096, np.uint8)
    n = po.lib().o_viterbi_decode(C.byref(c), 768, packed.ctypes.data_as(C.c_void_p), len(packed),
                                  out.ctypes.data_as(C.c_void_p))
    assert n == len(want) > 0
    assert (out[:n] == want).all()


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_chain_taps(po, name):
    _, const, cr, mode, nsf, seed, lead, snr = CASES[name]
    e = TAPS[name]
    c, iq = mg.make_case(const, cr, mode, nsf, seed, lead, snr)
    assert len(iq) == e["n_samples"] and mg.sha(iq) == e["iq_sha256"], "seeded generator drifted"
    o = po.rx(c, iq, snr_db=30.0 if snr is None else snr, want=mg.INT_TAPS)
    assert o["n_acquired"] == e["n_acquired"] and o["first_out_symbol"] == e["first_out_symbol"]
    assert mg.sha(o["cp_start"].astype(np.int32)) == e["cp_start_sha256"]
    assert mg.sha(o["sym_index"].astype(np.int32)) == e["sym_index_sha256"]
    for t in mg.INT_TAPS:
        assert o[t].size == e["taps"][t]["n"], t
        assert mg.sha(o[t]) == e["taps"][t]["sha256"], t
    assert (o["rs_fail"], o["rs_corr"]) == (e["rs_fail"], e["rs_corr"])


CH_CASES = {c[0]: c for c in mg.CHANNEL_CASES}


@pytest.mark.parametrize("name", sorted(CH_CASES))
def test_oracle_channel_taps(po, name):
    """round 3: carrier offsets, echoes, lock losses (bit-reproducible channel model, oracle/pyoracle.py::channel_exact)"""
    _, const, cr, mode, nsf, seed, chan = CH_CASES[name]
    e = TAPS[name]
    c, iq = mg.make_channel_case(const, cr, mode, nsf, seed, chan)
    assert len(iq) == e["n_samples"] and mg.sha(iq) == e["iq_sha256"], "seeded generator / channel model drifted"
    o = po.rx(c, iq, snr_db=30.0 if e["snr_db"] is None else e["snr_db"], want=mg.INT_TAPS)
    assert o["n_acquired"] == e["n_acquired"] and o["first_out_symbol"] == e["first_out_symbol"]
    assert [list(x) for x in o["lock_periods"]] == e["lock_periods"]
    for k in ("cp_start", "sym_index", "freq_offset"):
        assert mg.sha(o[k].astype(np.int32)) == e[k + "_sha256"], k
    for t in mg.INT_TAPS:
        assert o[t].size == e["taps"][t]["n"], t
        assert mg.sha(o[t]) == e["taps"][t]["sha256"], t
    assert (o["rs_fail"], o["rs_corr"]) == (e["rs_fail"], e["rs_corr"])


def test_oracle_rs_words(po):
    s = np.load(os.path.join(G, "chain_slices.npz"))
    L = po.lib()
    rs = po.RS()
    L.o_rs_init(C.byref(rs))
    words = np.ascontiguousarray(s["rs_in"])
    for compat in (0, 1):
        o = np.zeros((16, 188), np.uint8)
        nf, nc = C.c_int(), C.c_int()
        L.o_rs_dec_block(C.byref(rs), words.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p),
                         C.c_size_t(16), compat, C.byref(nf), C.byref(nc))
        assert (o == s["rs_out_compat%d" % compat]).all()
        assert nf.value == TAPS["rs_words_compat%d" % compat]["fail"]
    # <= 8 symbol errors are corrected by the correct decoder; the as-compiled reference (compat) drops the
    # correction at the lowest error location
    ok = s["rs_nerr"] <= 8
    assert (s["rs_out_compat0"][ok] != s["rs_out_compat1"][ok]).any()


HIER = json.load(open(os.path.join(G, "hier_taps.json")))
HIER_CASES = {c[0]: c for c in mg.HIER_CASES}


@pytest.mark.parametrize("name", sorted(HIER_CASES))
def test_oracle_hierarchical_taps(po, name):
    """round 4: alpha = 1 / 2 / 4 constellations, the bit de-interleaver's two outputs, the decoder on output 0 (tests/golden/hier_taps.json)"""
    _, const, hier, cr, mode, nsf, seed, lead = HIER_CASES[name]
    e = HIER[name]
    c, iq = mg.make_hier_case(const, hier, cr, mode, nsf, seed, lead)
    assert len(iq) == e["n_samples"] and mg.sha(iq) == e["iq_sha256"], "seeded generator drifted"
    o = po.rx(c, iq, want=mg.HIER_TAPS)
    assert o["n_acquired"] == e["n_acquired"] and o["first_out_symbol"] == e["first_out_symbol"]
    for t in mg.HIER_TAPS:
        assert o[t].size == e["taps"][t]["n"] and mg.sha(o[t]) == e["taps"][t]["sha256"], t
