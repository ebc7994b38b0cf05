"""The Viterbi stage is the reference's ONE streaming decoder (lib/viterbi_decoder_impl.cc:192-324, lib/d_viterbi.c:680-735) on every input, by construction.

The chunk decoders each start `viterbi_warm_windows` windows early from all-zero metrics; a chunk equals the streaming decoder from the window on at which its state
is its predecessor's, and whether that lies inside the warm-up depends on the INPUT (tools/hier_warmup.py): always on a stream the code can cope with,
not at about one chunk start in a few hundred on a collapsed channel (pre-Viterbi bit error rate 6 % at rate 7/8) or on the hierarchical modes' degenerate decoder
input.  Since round 6 every launch PROVES its chunks (own state == predecessor's state at the chunk's first window) and decodes the unproven ones again from the
predecessor's state (dvbt_rx_params.viterbi_verify, k_viterbi3.hpp): these tests hold the result -- 0 differing bytes with default parameters -- against the streaming
decoder (oracle/o_viterbi.c, pinned to the reference's own kernels) over the chain's OWN decoder input, and show with viterbi_verify = -1 / 2 what the plain chunk
decoders did before.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gr_dvbt_amd
    assert gr_dvbt_amd.device_count() > 0, "GPU tests need a GPU; the product path has no fallback"
    return gr_dvbt_amd


def test_warm_up_parameter_is_validated(g):
    for bad in (1, 24, 47, 100, 1176, -24):
        with pytest.raises(RuntimeError):
            g.Rx(g.QAM16, g.C1_2, g.T2k, max_samples=1 << 20, viterbi_warm_windows=bad)
    with pytest.raises(RuntimeError):
        g.Rx(g.QAM16, g.C1_2, g.T2k, max_samples=1 << 20, viterbi_warm_windows=144, soft_decision=1)
    g.Rx(g.QAM16, g.C1_2, g.T2k, max_samples=1 << 20, viterbi_warm_windows=48).close()
    g.Rx(g.QAM16, g.C1_2, g.T2k, max_samples=1 << 20, viterbi_warm_windows=1152).close()


@pytest.mark.parametrize("warm", [48, 144, 288])
def test_clean_stream_is_the_oracle_at_any_warm_up(po, g, warm):
    """the other instantiation of the kernel (warm-up from the parameters) on BASELINE config 2's mode and on 8k QAM64 7/8: every byte behind the decoder"""
    for const, cr, mode, nsf in ((g.QAM16, g.C1_2, g.T2k, 3), (g.QAM64, g.C7_8, g.T8k, 2)):
        c = po.cfg(const, cr, mode)
        ibits = c.payload * c.m * c.k // c.n
        iq = po.tx(c, po.make_ts((272 * ibits * nsf) // (204 * 8), 21), lead_in=900, tail=3 * c.N)
        o = po.rx(c, iq, want=("vit", "rs", "ts"))
        rx = g.Rx(const, cr, mode, max_samples=len(iq), taps=True, viterbi_warm_windows=warm)
        rx.run(iq)
        for name, tap in (("vit", g.TAP_VITERBI), ("rs", g.TAP_RS), ("ts", g.TAP_TS)):
            a, b = rx.tap(tap).reshape(-1), o[name].reshape(-1)
            assert a.size == b.size > 0 and (a == b).all(), (name, warm)
        rx.close()


def _streaming_reference(po, c, bd):
    """the streaming decoder (oracle/o_viterbi.c, and -- where oracle/_ref is built -- the reference's own lib/d_viterbi.c kernels) over the decoder input bd"""
    import ctypes as C
    import os
    po.lib().o_viterbi_decode.restype = C.c_size_t
    bd = np.ascontiguousarray(bd.reshape(-1))
    ref = np.zeros(bd.size * c.m * c.k // (8 * c.n) + 64, np.uint8)
    n = po.lib().o_viterbi_decode(C.byref(c), 768, bd.ctypes.data_as(C.c_void_p), C.c_size_t(bd.size), ref.ctypes.data_as(C.c_void_p))
    if os.path.exists(po._REF):
        # the same loop around the REFERENCE's own kernels (oracle/_ref = lib/d_viterbi.c compiled unmodified; the prebuilt library travels to the GPU box):
        # the restatement is those kernels on this input too, so what is compared is reference execution
        po.lib().o_ref_viterbi_decode_n.restype = C.c_size_t
        po.lib().o_ref_viterbi_decode_n.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        d_nsym = 768 * c.n // c.m
        whole = (bd.size // d_nsym) * d_nsym                     # o_viterbi_decode takes whole blocks (viterbi_decoder_impl.cc:198)
        rk = np.zeros_like(ref)
        nk = po.lib().o_ref_viterbi_decode_n(po._REF.encode(), C.byref(c), bd.ctypes.data_as(C.c_void_p), C.c_size_t(whole), rk.ctypes.data_as(C.c_void_p))
        assert nk == n and (rk[:n] == ref[:n]).all()
    return ref[:n]


def test_collapsed_channel_is_the_streaming_decoder_with_default_parameters(po, g):
    """2k QAM64 7/8 at 16 dB (ofdm_sym_acquisition's snr = the channel's: the CP lock holds, one lock period): pre-Viterbi bit error rate 6 %, 1.35 MB through the
    decoder in ~5,000 chunks.  The reference here is the streaming decoder over the chain's OWN decoder input (the BITDEINT tap: under noise a hard decision within
    float rounding of a boundary may differ from the oracle's, DESIGN.md 7 -- that is not this test's subject).  The plain chunk decoders (viterbi_verify = -1, what the
    library ran until round 5) differ at a handful of chunk starts on this input; the default -- proof + repair -- is the streaming decoder byte for byte, at the default
    warm-up and at a shorter and a longer one, and so is the sequential pass alone (viterbi_verify = 3, the test hook that leaves every repair to it)."""
    c = po.cfg(po.QAM64, po.C7_8, po.T2k)
    ibits = c.payload * c.m * c.k // c.n
    iq = po.tx(c, po.make_ts((272 * ibits * 6) // (204 * 8), 5), lead_in=500, tail=3 * c.N)
    iq = po.channel(iq, c.N, snr_db=16, seed=5)
    o = po.rx(c, iq, snr_db=16.0, want=("bitdeint", "vit"))
    assert len(o["lock_periods"]) == 1 and len(o["vit"]) > 1000000
    res = {}
    for name, kw in (("plain", dict(viterbi_verify=-1)), ("default", dict()), ("default+count", dict(viterbi_verify=1)), ("warm 48", dict(viterbi_warm_windows=48, viterbi_verify=1)),
                     ("warm 288", dict(viterbi_warm_windows=288, viterbi_verify=1)), ("sequential", dict(viterbi_verify=3)), ("hand-over", dict(viterbi_verify=4))):
        rx = g.Rx(po.QAM64, po.C7_8, po.T2k, max_samples=len(iq), taps=True, snr_db=16.0, **kw)
        rep = rx.run(iq)
        assert rep.first_out_symbol == o["first_out_symbol"] and rep.n_lock_periods == 1
        bd = rx.tap(g.TAP_BITDEINT)
        assert bd.size == o["bitdeint"].size
        ref = _streaming_reference(po, c, bd)
        v = rx.tap(g.TAP_VITERBI)
        assert len(v) == len(ref) == len(o["vit"])
        res[name] = (int((v != ref).sum()), rx.viterbi_proof() if name != "plain" else None)
        rx.close()
    print("collapsed channel, (Viterbi bytes that differ from the streaming decoder over the same input, the passes' counters):", res)
    assert 0 < res["plain"][0] < len(o["vit"]) // 200, res               # the plain chunk decoders: wrong at a small fraction of the chunk starts
    for name in ("default", "default+count", "warm 48", "warm 288", "sequential", "hand-over"):
        assert res[name][0] == 0, (name, res)
    p = res["default+count"][1]
    assert p["chunks"] > 1000 and 0 < p["decoded_again"] < p["chunks"] // 20 and p["not_proven"] == 0, p
    assert res["default"][1]["not_proven"] == -1 and res["default"][1]["decoded_again"] == p["decoded_again"]      # (no final check by default)
    assert res["warm 48"][1]["decoded_again"] > p["decoded_again"] > res["warm 288"][1]["decoded_again"] == 0, res   # the warm-up moves the number of repairs, not the bytes
    q = res["sequential"][1]
    assert q["sequential"] >= q["decoded_again"] == p["decoded_again"] and q["not_proven"] == 0, q
    # viterbi_verify = 4: every repaired chunk flags the chunk behind it with the state it arrived in -- the hand-over the sequential pass exists for, which no input seen so far
    # reaches by itself: that pass then decodes the flagged chunks (one each: it arrives in the state they were decoded from), and the bytes stay the streaming decoder's
    q = res["hand-over"][1]
    assert q["decoded_again"] == p["decoded_again"] and q["sequential"] >= 1 and q["not_proven"] == 0, q


@pytest.mark.parametrize("const,hier,mode,cr", [(2, 2, 0, 2), (2, 3, 1, 4), (1, 2, 0, 0)], ids=["2k QAM64 alpha 2 2/3", "8k QAM64 alpha 4 7/8", "2k QAM16 alpha 2 1/2"])
def test_hierarchical_modes_chunked_decoder(po, g, const, hier, mode, cr):
    """the hierarchical modes (ONE decoder on one wavefront until round 5): the chunk decoders with default parameters equal the oracle on both priority streams -- their degenerate
    decoder input (two thirds constant zeros) leaves some chunks unproven at 72 windows, the repair pass decodes those again"""
    c = po.cfg(const, cr, mode, hierarchy=hier)
    ibits = c.payload * c.m * c.k // c.n
    iq = po.tx(c, po.make_ts((272 * ibits * 3) // (204 * 8), 14), lead_in=700, tail=3 * c.N)
    o = po.rx(c, iq, want=("bitdeint", "bitdeint_lp", "vit", "rs"))
    rx = g.Rx(const, cr, mode, max_samples=len(iq), hierarchy=hier, taps=True, viterbi_verify=1)
    rep = rx.run(iq)
    assert rep.first_out_symbol == o["first_out_symbol"] >= 0
    for name, tap in (("bitdeint", g.TAP_BITDEINT), ("vit", g.TAP_VITERBI), ("rs", g.TAP_RS)):
        a, b = rx.tap(tap).reshape(-1), o[name].reshape(-1)
        assert a.size == b.size > 0 and (a == b).all(), name
    pr = rx.viterbi_proof()
    print("hierarchical, HP stream:", pr)
    assert pr["chunks"] >= 2 and pr["not_proven"] == 0
    rx.close()
    import ctypes as C
    lp = o["bitdeint_lp"].reshape(-1)
    ref = np.zeros(len(lp) * c.m * c.k // (8 * c.n) + 64, np.uint8)
    po.lib().o_viterbi_decode.restype = C.c_size_t
    n = po.lib().o_viterbi_decode(C.byref(c), 768, lp.ctypes.data_as(C.c_void_p), C.c_size_t(len(lp)), ref.ctypes.data_as(C.c_void_p))
    rx = g.Rx(const, cr, mode, max_samples=len(iq), hierarchy=hier, taps=True, hier_stream=1)
    rx.run(iq)
    v = rx.tap(g.TAP_VITERBI)
    assert len(v) == n and (v == ref[:n]).all()
    rx.close()


def test_viterbi_proof_counts(po, g):
    """dvbt_rx_params.viterbi_verify: every chunk decoder leaves its state at its chunk's first window, its predecessor (the streaming decoder there, by induction) its own,
    a checker compares -- equal states make equal decisions (tests/test_viterbi_boundary_proof_model.py holds the criterion on the CPU).  A clean stream: every chunk proven by
    the warm-up alone, the oracle's bytes.  The collapsed channel of the test above with viterbi_verify = 2 (proof only, nothing decoded again): some chunks are NOT proven and
    bytes do differ from the streaming decoder; nowhere: all chunks proven and a byte different."""
    # clean, the headline's mode
    c = po.cfg(po.QAM64, po.C7_8, po.T8k)
    ibits = c.payload * c.m * c.k // c.n
    iq = po.tx(c, po.make_ts((272 * ibits * 2) // (204 * 8), 21), lead_in=900, tail=3 * c.N)
    o = po.rx(c, iq, want=("vit", "ts"))
    rx = g.Rx(po.QAM64, po.C7_8, po.T8k, max_samples=len(iq), taps=True, viterbi_verify=1)
    rx.run(iq)
    chunks, unproven = rx.viterbi_check()
    pr = rx.viterbi_proof()
    assert chunks >= 2 and unproven == 0 and pr["decoded_again"] == 0 and pr["sequential"] == 0, (chunks, unproven, pr)
    for name, tap in (("vit", g.TAP_VITERBI), ("ts", g.TAP_TS)):
        a, b = rx.tap(tap).reshape(-1), o[name].reshape(-1)
        assert a.size == b.size > 0 and (a == b).all(), name
    rx.close()
    with pytest.raises(RuntimeError):                                   # the default runs no final check: nothing to report there ...
        h = g.Rx(po.QAM64, po.C7_8, po.T8k, max_samples=len(iq)); h.run(iq); h.viterbi_check()
    with pytest.raises(RuntimeError):                                   # ... and the plain chunk decoders keep no states at all
        h = g.Rx(po.QAM64, po.C7_8, po.T8k, max_samples=len(iq), viterbi_verify=-1); h.run(iq); h.viterbi_proof()
    with pytest.raises(RuntimeError):
        g.Rx(po.QAM64, po.C7_8, po.T8k, max_samples=len(iq), viterbi_verify=5)
    # the collapsed channel, proof only
    c = po.cfg(po.QAM64, po.C7_8, po.T2k)
    ibits = c.payload * c.m * c.k // c.n
    iq = po.channel(po.tx(c, po.make_ts((272 * ibits * 6) // (204 * 8), 5), lead_in=500, tail=3 * c.N), c.N, snr_db=16, seed=5)
    res = {}
    for warm in (0, 288):
        rx = g.Rx(po.QAM64, po.C7_8, po.T2k, max_samples=len(iq), taps=True, snr_db=16.0, viterbi_warm_windows=warm, viterbi_verify=2)
        rx.run(iq)
        chunks, unproven = rx.viterbi_check()
        ref = _streaming_reference(po, c, rx.tap(g.TAP_BITDEINT))
        v = rx.tap(g.TAP_VITERBI)
        assert len(v) == len(ref)
        res[warm] = (chunks, unproven, int((v != ref).sum()))
        rx.close()
    print("collapsed channel, proof only: (chunks, chunks not proven, bytes that differ from the streaming decoder) by warm-up (0 = the default 72):", res)
    for chunks, unproven, diff in res.values():
        assert chunks > 1000 and (unproven > 0 or diff == 0)            # all proven => no byte differs
    assert res[288][1] == 0 and res[288][2] == 0, res
    assert 0 < res[0][1] < res[0][0] // 20 and res[0][2] > 0, res      # the warm-up alone on this input: a few chunks cannot be proven, and bytes do differ


def test_viterbi_block_is_the_streaming_decoder(po, g):
    """the single block (dvbt_viterbi_decoder_*, chunks of 264 bytes) on a decoder input with 6 % bit errors at rate 7/8, fed in calls of 40 blocks: every call's launch proves
    and repairs its chunks and the decoder's state is carried from call to call -- 0 bytes differ from the streaming decoder at any warm-up, and some chunks were decoded again"""
    import ctypes as C
    c = po.cfg(po.QAM64, po.C7_8, po.T2k)
    ibits = c.payload * c.m * c.k // c.n
    iq = po.tx(c, po.make_ts((272 * ibits * 4) // (204 * 8), 5), lead_in=500, tail=3 * c.N)
    vin = po.rx(c, iq, want=("bitdeint",))["bitdeint"].reshape(-1).copy()
    rng = np.random.RandomState(3)
    for b in range(c.m):
        vin ^= (rng.rand(len(vin)) < 0.06).astype(np.uint8) << b
    d_nsym, d_nout = 768 * c.n // c.m, 768 * c.k // 8
    nblocks = len(vin) // d_nsym
    vin = np.ascontiguousarray(vin[:nblocks * d_nsym])
    ref = _streaming_reference(po, c, vin)
    L = g.lib()
    L.dvbt_viterbi_decoder_set_warm_windows.argtypes = [C.c_void_p, C.c_int]
    res = {}
    for warm, per_call in ((0, 40), (48, 1), (288, 7)):
        b = g.Block("viterbi_decoder", 2, 0, 4, 768, 0, -1)
        assert L.dvbt_viterbi_decoder_set_warm_windows(b.h, 600) < 0 and L.dvbt_viterbi_decoder_set_warm_windows(b.h, warm) == 0
        outs, pos, first = [], 0, True
        while pos < nblocks:
            nb = min(per_call, nblocks - pos)
            o = np.zeros(nb * d_nout, np.uint8)
            r, cons, _ = b.work(nb * d_nout, nb * d_nsym, vin[pos * d_nsym:(pos + nb) * d_nsym], o, tags=[(0, g.TAG_SUPERFRAME_START, 0xaa)] if first else [])
            assert cons == nb * d_nsym
            outs.append(o[:r]); pos += nb; first = False
        out = np.concatenate(outs)
        assert len(out) == len(ref)
        res[warm] = (int((out != ref).sum()), b.viterbi_proof())
        b.close()
    print("viterbi block, 6 % bit errors at rate 7/8: (bytes that differ from the streaming decoder, the passes' counters) by warm-up (0 = the default 72):", res, "of", len(ref))
    for warm, (diff, pr) in res.items():
        assert diff == 0, res
    assert res[0][1]["decoded_again"] > 0 and res[48][1]["decoded_again"] > 0, res


def test_streaming_entry_repairs_its_chunks_too(po, g):
    """dvbt_rx_stream_* on the collapsed channel (2k QAM64 7/8 at 16 dB, pieces of two superframes, ragged pushes): the stream's chains prove and repair every launch like any
    handle -- chunks WERE decoded again on the way (dvbt_rx_stream_viterbi_proof sums the passes' counters over the stream's launches).  The TS against the single chain's: the same
    length, and the same bytes but for what failed RS words pass through -- every RS word fails on this channel, and a piece's chain starts its derotation phase afresh 76 symbols in
    front of its boundary, 1e-5 rad beside the long-running chain's: a hard decision within float rounding of a boundary, not the decoder."""
    c = po.cfg(po.QAM64, po.C7_8, po.T2k)
    ibits = c.payload * c.m * c.k // c.n
    iq = po.channel(po.tx(c, po.make_ts((272 * ibits * 9) // (204 * 8), 5), lead_in=500, tail=3 * c.N), c.N, snr_db=16, seed=5)
    rx = g.Rx(po.QAM64, po.C7_8, po.T2k, max_samples=len(iq), snr_db=16.0)
    rep = rx.run(iq)
    single = rx.tap(g.TAP_TS).copy()
    rx.close()
    assert rep.n_lock_periods == 1 and rep.n_viterbi_bytes > 2000000 and len(single) > 100000      # (the descrambler finds its NSYNC in a fraction of the garbage only)
    st = g.RxStream(po.QAM64, po.C7_8, po.T2k, segment_superframes=2, snr_db=16.0)
    out = []
    for a in range(0, len(iq), 123457):
        st.push(iq[a:a + 123457]); out.append(st.pull())
    st.finish(); out.append(st.pull())
    pr = st.viterbi_proof()
    st.close()
    ts = np.concatenate(out)
    ndiff = int((ts != single).sum()) if len(ts) == len(single) else -1
    print("streaming entry on the collapsed channel:", pr, "TS bytes", len(ts), "differing from the single chain's:", ndiff)
    assert pr["chunks"] > 1000 and pr["decoded_again"] > 0
    assert len(ts) == len(single) and 0 <= ndiff <= len(single) // 200
