"""Soft-decision mode (dvbt_rx_params.soft_decision = 1; SURVEY 8f row 4, second half; gr-dvbt's TODO.txt:25-26).  The reference decodes hard decisions only:
there is no oracle for this mode and no parity claim.  It is validated against the hard path on identical samples: the same TS, byte for byte, on clean
loopbacks (and on noisy ones as long as the hard path still decodes everything), and a working decoder where the hard path has collapsed -- at 1 dB and at
2 dB below the SNR at which the hard path's packet error rate passes 50 %."""
import numpy as np
import pytest

import gr_dvbt_amd as g

pytestmark = pytest.mark.gpu


def both(po, const, cr, mode, nsf, snr=None, echoes=(), seed=9, rx_snr=None):
    c = po.cfg(const, cr, mode)
    clean = po.stream_slice(c, nsf, seed)
    iq = po.channel(clean, c.N, echoes=echoes, snr_db=snr, seed=5) if (snr is not None or echoes) else clean
    sent = {bytes(p) for p in po.stream_ts(c, 0, nsf, seed).reshape(-1, 188)}
    out = []
    for soft in (0, 1):
        rx = g.Rx(const, cr, mode, max_samples=len(iq), snr_db=rx_snr if rx_snr is not None else 30.0 if snr is None else snr, soft_decision=soft)
        rep = rx.run(iq)
        ts = rx.tap(g.TAP_TS).copy()
        pk = ts.reshape(-1, 188)
        per = 1.0 if len(pk) == 0 else 1.0 - sum(1 for p in pk if bytes(p) in sent) / len(pk)
        out.append({"ts": ts, "rep": rep, "per": per, "fail": rep.rs_fail_words, "corr": rep.rs_corrected_symbols, "periods": rep.n_lock_periods})
        rx.close()
    return out


@pytest.mark.parametrize("const,cr,mode,nsf", [(g.QAM16, g.C1_2, g.T2k, 3), (g.QAM64, g.C7_8, g.T8k, 2), (g.QPSK, g.C2_3, g.T8k, 2), (g.QAM64, g.C3_4, g.T2k, 3)])
def test_clean_loopback_equals_the_hard_path(po, const, cr, mode, nsf):
    hard, soft = both(po, const, cr, mode, nsf)
    assert len(soft["ts"]) == len(hard["ts"]) > 0 and (soft["ts"] == hard["ts"]).all()
    assert soft["fail"] == hard["fail"] == 11 and soft["corr"] == 0


@pytest.mark.parametrize("const,cr,mode,nsf,snr_ok,snr_hard_dead", [
    (g.QAM16, g.C1_2, g.T2k, 4, 12.0, 10.0),          # hard: 787 corrected symbols at 12 dB, PER 0.91 at 10 dB
    (g.QAM64, g.C7_8, g.T8k, 2, 22.0, 20.0),          # hard: 12,558 corrected at 22 dB, PER 1.0 at 20 dB
    (g.QAM64, g.C2_3, g.T2k, 4, 20.0, 17.0),
])
def test_noise_soft_decodes_where_hard_has_collapsed(po, const, cr, mode, nsf, snr_ok, snr_hard_dead):
    hard, soft = both(po, const, cr, mode, nsf, snr=snr_ok)
    # both still deliver every packet: identical TS; the soft decoder leaves the RS decoder (much) less to do
    assert hard["per"] < 0.02 and (soft["ts"] == hard["ts"]).all()
    assert hard["corr"] > 100 and soft["corr"] <= hard["corr"] // 10
    hard, soft = both(po, const, cr, mode, nsf, snr=snr_hard_dead)
    assert hard["per"] > 0.5, "the point is meant to lie below the hard path's waterfall"
    assert soft["periods"] == hard["periods"] == 1
    assert soft["per"] < 0.01 and soft["fail"] <= 11 + 2


def test_frequency_selective_channel(po):
    """an echo at -10 dB: carriers in the fades carry less weight (channel state information in the LLRs)"""
    hard, soft = both(po, g.QAM16, g.C3_4, g.T2k, 4, snr=16.0, echoes=((19, 0.3),))
    assert soft["periods"] == hard["periods"] and len(soft["ts"]) == len(hard["ts"]) > 0
    assert soft["corr"] < hard["corr"] and soft["per"] <= hard["per"]


def test_deep_echo_below_the_hard_waterfall(po):
    """an echo at -6 dB inside the guard interval (notches 9.5 dB below the peaks), 16 dB of noise: the hard path has collapsed (PER 0.71 over 8 superframes,
    tools/soft_gain.py echo), the soft path with its channel-state weights delivers every packet (3 dB of gain on this channel against 2.3 dB on a flat one).
    ofdm_sym_acquisition's snr parameter at 10 dB: at the demo flowgraphs' 30 dB the reference's peak detector drops the lock on this channel."""
    hard, soft = both(po, g.QAM16, g.C3_4, g.T2k, 8, snr=16.0, echoes=((20, 0.5 + 0j),), rx_snr=10.0)
    assert soft["periods"] == hard["periods"] == 1
    assert hard["per"] > 0.3 and soft["per"] < 0.01


@pytest.mark.parametrize("const,cr,mode,nsf,seg_sf,call", [(g.QAM16, g.C1_2, g.T2k, 9, 2, 4 * 2112), (g.QAM64, g.C7_8, g.T8k, 6, 1, 33 * 8448 + 5)])
def test_soft_mode_through_the_streaming_entry(po, const, cr, mode, nsf, seg_sf, call):
    """dvbt_rx_stream_* with soft_decision = 1: pieces that continue a cut stream (dvbt_rx_set_cut: sizes in stream coordinates) go through the soft decoder
    too; on a clean stream the TS pulled is the hard chain's over the whole stream"""
    c = po.cfg(const, cr, mode)
    iq = po.stream_slice(c, nsf, 9)
    ref = po.rx(c, iq, want=("ts",))["ts"].copy()                  # the oracle's chain over the whole stream (a clean stream: soft and hard decisions decode the same bytes)
    st = g.RxStream(const, cr, mode, segment_superframes=seg_sf, soft_decision=1)
    out = []
    for pos in range(0, len(iq), call):
        st.push(iq[pos:pos + call])
        out.append(st.pull())
    st.finish()
    out.append(st.pull())
    info = st.info()
    st.close()
    ts = np.concatenate(out)
    assert info.status & ~2 == 0 and len(ts) == len(ref) > 0 and (ts == ref).all()


def _sharded(po, iq, world, call, **kw):
    """world stream objects in one process (one GPU), every rank pushed the same stream; the ranks' chunks ordered by their packet index"""
    ranks = [g.RxStream(rank=r, world=world, **kw) for r in range(world)]
    chunks = []
    for pos in range(0, len(iq), call):
        for st in ranks:
            st.push(iq[pos:pos + call])
        for r, st in enumerate(ranks):
            chunks += [(fp, r, b) for fp, b in st.pull_chunks()]
    for r, st in enumerate(ranks):
        st.finish()
        chunks += [(fp, r, b) for fp, b in st.pull_chunks()]
    infos = [st.info() for st in ranks]
    for st in ranks:
        st.close()
    chunks.sort(key=lambda t: t[0])
    at = chunks[0][0]
    for fp, r, b in chunks:                                        # contiguous, no packet twice
        assert fp == at, (fp, at, r)
        at += len(b) // 188
    return np.concatenate([b for _, _, b in chunks]), infos, {r for _, r, _ in chunks}


def test_soft_mode_sharded(po):
    """soft decisions with rank / world: the pieces of a sharded stream go through the soft decoder on every rank; the chunks in packet order are the oracle's TS"""
    const, cr, mode = g.QAM64, g.C7_8, g.T8k
    c = po.cfg(const, cr, mode)
    iq = po.stream_slice(c, 8, 9)
    ref = po.rx(c, iq, want=("ts",))["ts"]
    ts, infos, who = _sharded(po, iq, 2, 40 * 8448 + 3, constellation=const, code_rate=cr, mode=mode, segment_superframes=1, soft_decision=1)
    assert all(i.status & ~2 == 0 for i in infos) and who == {0, 1}
    assert len(ts) == len(ref) > 0 and (ts == ref).all()


# ---------------------------------------------------------------- the soft path against an independent model (oracle/o_soft.c: a MODEL of the kernels' specification,
# not of the reference, which has no soft path).  The GPU's own EQ / CSI taps go into the model's demapper, the GPU's soft values into the model's decoder:
# soft values and decoded bytes must be IDENTICAL.  A wrong weight, tie rule, depuncturing phase or traceback start would cost tenths of a dB that no
# packet-error threshold notices; here it is a differing byte.
def _model_check(po, const, cr, mode, nsf, snr, echoes=(), seed=9, optional=False):
    import ctypes as C
    c = po.cfg(const, cr, mode)
    clean = po.stream_slice(c, nsf, seed)
    iq = po.channel(clean, c.N, echoes=echoes, snr_db=snr, seed=5)
    import torch
    dev = torch.from_numpy(iq.view(np.float32)).cuda()
    torch.cuda.synchronize()
    rx = g.Rx(const, cr, mode, max_samples=len(iq), snr_db=snr, soft_decision=1)
    rx.enqueue_device(dev.data_ptr(), len(iq))                       # the asynchronous entry: the stream's first lock period, one launch of every kernel
    rep = rx.finish()
    if optional and not (rep.first_out_symbol >= 0 and rep.n_out_symbols >= 272):
        rx.close()
        return None
    assert rep.first_out_symbol >= 0 and rep.n_out_symbols >= 272, "the point is meant to be one the reference's tracker holds for a superframe at least"
    eq, csi, soft, vit = rx.tap(g.TAP_EQ), rx.tap(g.TAP_CSI), rx.tap(g.TAP_SOFT), rx.tap(g.TAP_VITERBI).copy()
    sidx = rx.tap(g.TAP_SYMBOL_INDEX)
    rx.close()
    n_out, P, m = rep.n_out_symbols, c.payload, c.m
    assert eq.shape == (n_out, P) and csi.shape == (n_out, P) and soft.shape == (n_out, P * m)
    assert (csi > 0).all() and np.isfinite(csi).all()
    parity = np.ascontiguousarray(sidx[rep.first_out_symbol:rep.first_out_symbol + n_out] & 1, dtype=np.int32)
    L = po.lib()
    # ---- the demapper
    model_soft = np.zeros((n_out, P * m), np.int8)
    L.o_soft_demap.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.o_soft_demap(C.byref(c), np.ascontiguousarray(eq).ctypes.data, np.ascontiguousarray(csi).ctypes.data, parity.ctypes.data, n_out, model_soft.ctypes.data)
    diff = soft.astype(np.int32) - model_soft.astype(np.int32)
    assert np.abs(diff).max() == 0, (int(np.abs(diff).max()), int((diff != 0).sum()))
    hist = np.bincount(np.abs(soft.reshape(-1).astype(np.int32)), minlength=32)
    assert hist[31] < 0.9 * soft.size and hist[:8].sum() > 0.01 * soft.size        # the values use their range: neither all clamped nor all confident
    # ---- the decoder
    ntb = {0: 5, 1: 9, 2: 10, 3: 15, 4: 24}[cr]
    total_steps = (rep.n_viterbi_bytes + ntb) * 8
    n_soft = total_steps * c.n // c.k
    Lc = c.N + c.cp
    ncalls = (len(iq) - (2 * c.N + c.cp + 16)) // Lc + 1
    max_vit = ncalls * P * m * c.k // (8 * c.n) + 1
    B, nsteps = C.c_int(), C.c_int()
    L.o_soft_plan.argtypes = [C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]
    L.o_soft_plan(max_vit, ntb, C.byref(B), C.byref(nsteps))
    out = np.zeros(rep.n_viterbi_bytes + 64, np.uint8)
    L.o_soft_viterbi.restype = C.c_longlong
    L.o_soft_viterbi.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_void_p]
    flat = np.ascontiguousarray(soft.reshape(-1))
    n = L.o_soft_viterbi(C.byref(c), flat.ctypes.data, n_soft, total_steps, B.value, nsteps.value, out.ctypes.data)
    assert n == rep.n_viterbi_bytes == len(vit)
    bad = np.flatnonzero(out[:n] != vit)
    assert len(bad) == 0, (len(bad), bad[:8], B.value, nsteps.value)
    return rep


@pytest.mark.parametrize("const,cr,mode,nsf,snr", [(g.QAM16, g.C1_2, g.T2k, 4, 10.0), (g.QAM64, g.C7_8, g.T8k, 2, 20.0), (g.QPSK, g.C2_3, g.T8k, 2, 30.0), (g.QAM64, g.C2_3, g.T2k, 4, 17.0)],
                         ids=["2k QAM16 1/2 at 10 dB (hard path dead)", "8k QAM64 7/8 at 20 dB (hard path dead)", "8k QPSK 2/3 clean", "2k QAM64 2/3 at 17 dB (hard path dead)"])
def test_soft_values_and_decoded_bytes_equal_the_model(po, const, cr, mode, nsf, snr):
    rep = _model_check(po, const, cr, mode, nsf, snr)
    if snr < 25:
        assert rep.rs_corrected_symbols > 0 or rep.rs_fail_words > 11          # the decoder had errors to make: the comparison is not of two clean streams


def test_soft_values_and_decoded_bytes_equal_the_model_on_an_echo_channel(po):
    """an echo of -10 dB: the channel-state weights vary by 10 dB across the carriers (the cap at 4 and the mean both matter).  The asynchronous entry decodes the
    stream's FIRST lock period: the noise level is raised until the reference's tracker holds that one for a superframe"""
    for snr in (16.0, 18.0, 20.0, 24.0):
        rep = _model_check(po, g.QAM16, g.C3_4, g.T2k, 4, snr, echoes=((19, 0.3),), optional=True)
        if rep is not None:
            return
    pytest.fail("no noise level at which the first lock period delivers a superframe")
