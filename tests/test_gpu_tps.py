"""TPS auto-detection (the reference author's TODO.txt:25-28): the report carries the transmission parameters
the stream itself signals -- the TPS word of a BCH-valid frame (reference_signals_impl.cc:385-425) decoded per the layout of
format_tps_data (:883-916) -- and a handle configured differently is told so."""
import ctypes as C
import numpy as np
import pytest

import gr_dvbt_amd as g

pytestmark = pytest.mark.gpu

# the five BASELINE configurations (1 and 2 share their parameters) + one more rate
CONFIGS = [(g.QAM16, g.C1_2, g.T2k), (g.QAM64, g.C7_8, g.T8k), (g.QPSK, g.C7_8, g.T8k), (g.QAM64, g.C3_4, g.T2k)]


@pytest.mark.parametrize("const,cr,mode", CONFIGS)
def test_report_carries_the_signalled_parameters(po, const, cr, mode):
    c = po.cfg(const, cr, mode)
    iq = po.stream_slice(c, 2, 3)
    rx = g.Rx(const, cr, mode, max_samples=len(iq))
    rep = rx.run(iq)
    rx.close()
    assert rep.tps_valid == 1 and rep.tps_mismatch == 0 and not rep.status & 16
    assert (rep.tps_constellation, rep.tps_hierarchy, rep.tps_code_rate_hp, rep.tps_code_rate_lp) == (const, g.NH, cr, cr)
    assert (rep.tps_guard_interval, rep.tps_transmission_mode, rep.tps_cell_id) == (g.G1_32, mode, 0)
    assert rep.tps_length_indicator == 0x17                       # no cell id (:898-901)
    # the raw word against the transmitter's (o_tps_format, pinned by the Appendix F known-answer test): bits s17-s22, s25-s53
    wk = np.zeros(c.Kmax + 1, np.int8)
    po.lib().o_prbs_wk(C.byref(c), wk.ctypes.data_as(C.c_void_p))
    tps = np.zeros(68, np.uint8)
    po.lib().o_tps_format(C.byref(c), 1, wk.ctypes.data_as(C.c_void_p), tps.ctypes.data_as(C.c_void_p))
    want = sum(int(tps[i]) << i for i in list(range(17, 23)) + list(range(25, 54)))
    assert rep.tps_bits == want


def test_a_wrongly_configured_handle_is_told(po):
    """8k QAM64 7/8 on the air, the handle set up for 8k QAM16 2/3: the front end is the same, so the TPS is received and decoded,
    and the report names what differs (constellation and code rate)."""
    c = po.cfg(g.QAM64, g.C7_8, g.T8k)
    iq = po.stream_slice(c, 2, 5)
    rx = g.Rx(g.QAM16, g.C2_3, g.T8k, max_samples=len(iq))
    rep = rx.run(iq)
    rx.close()
    assert rep.tps_valid == 1 and rep.status & 16
    assert rep.tps_mismatch == 1 | 4
    assert (rep.tps_constellation, rep.tps_code_rate_hp, rep.tps_transmission_mode) == (g.QAM64, g.C7_8, g.T8k)
    from gr_dvbt_amd import multi
    with pytest.raises(ValueError):                               # such a piece is not stitched
        multi.stitch_plan([multi.piece_meta(rep)], g.get_dims(g.QAM16, g.C2_3, g.T8k))


def test_no_signal_no_tps(po):
    rng = np.random.RandomState(1)
    iq = (1e-3 * (rng.randn(200000) + 1j * rng.randn(200000))).astype(np.complex64)
    rx = g.Rx(g.QAM16, g.C1_2, g.T2k, max_samples=len(iq))
    rep = rx.run(iq)
    rx.close()
    assert rep.tps_valid == 0 and rep.tps_bits == 0 and rep.tps_mismatch == 0


# ---------------------------------------------------------------- TPS auto-CONFIGURATION of the streaming entry (TODO.txt:28 "Autodetect transmission params")
def _stream(po, const, cr, mode, iq, call, hierarchy=g.NH, **kw):
    st = g.RxStream(const, cr, mode, segment_superframes=2, hierarchy=hierarchy, **kw)
    out = []
    for a in range(0, len(iq), call):
        st.push(iq[a:a + call])
        out.append(st.pull())
    st.finish()
    out.append(st.pull())
    info = st.info()
    st.close()
    return np.concatenate(out), info


@pytest.mark.parametrize("const,cr,mode", CONFIGS)
def test_auto_configured_stream_equals_the_configured_one(po, const, cr, mode):
    """constellation / hierarchy / code_rate = AUTO: the head of the stream is held back until a probe chain has read a BCH-valid TPS word, the chains are built for
    what it names, the head is replayed.  The TS must be the oracle's chain with the true parameters (= the configured stream's), for each BASELINE config."""
    c = po.cfg(const, cr, mode)
    iq = po.stream_slice(c, 7, 21)
    want = po.rx(c, iq, want=("ts",))["ts"]
    L = c.N + c.cp
    ts, info = _stream(po, g.AUTO, g.AUTO, mode, iq, 50 * L + 7, hierarchy=g.AUTO)
    assert info.auto_configured == 1 and (info.constellation, info.hierarchy, info.code_rate) == (const, g.NH, cr)
    assert info.status & ~2 == 0 and len(ts) == len(want) > 0 and (ts == want).all()
    ref, info2 = _stream(po, const, cr, mode, iq, 50 * L + 7)
    assert info2.auto_configured == 0 and (ref == ts).all()


def test_auto_only_the_code_rate_and_device_pushes(po):
    """one field AUTO, the others given; samples pushed from device memory; a head that arrives in 4-symbol calls"""
    import torch
    const, cr, mode = g.QAM64, g.C7_8, g.T8k
    c = po.cfg(const, cr, mode)
    iq = po.stream_slice(c, 5, 22)
    want = po.rx(c, iq, want=("ts",))["ts"]
    dev = torch.from_numpy(iq.view(np.float32)).cuda()
    torch.cuda.synchronize()
    st = g.RxStream(const, g.AUTO, mode, segment_superframes=1)
    out, step = [], 4 * (c.N + c.cp)
    for a in range(0, len(iq), step):
        st.push_device(dev.data_ptr() + 8 * a, min(step, len(iq) - a))
        out.append(st.pull())
    st.finish(); out.append(st.pull())
    info = st.info(); st.close()
    ts = np.concatenate(out)
    assert info.auto_configured == 1 and info.code_rate == cr and len(ts) == len(want) and (ts == want).all()


def test_auto_detects_a_hierarchical_transmission(po):
    c = po.cfg(g.QAM64, g.C2_3, g.T2k, hierarchy=g.ALPHA2)
    ibits = c.payload * c.m * c.k // c.n
    iq = po.tx(c, po.make_ts((272 * ibits * 3) // (204 * 8), 5), lead_in=500, tail=3 * c.N)
    ts, info = _stream(po, g.AUTO, g.AUTO, g.T2k, iq, 30000, hierarchy=g.AUTO)
    assert info.auto_configured == 1 and (info.constellation, info.hierarchy, info.code_rate) == (g.QAM64, g.ALPHA2, g.C2_3)


def test_auto_short_stream_and_no_signal(po):
    # a stream that ends before the look-ahead is full: finish() detects on what there is
    const, cr, mode = g.QAM16, g.C1_2, g.T2k
    c = po.cfg(const, cr, mode)
    iq = po.stream_slice(c, 3, 23)
    want = po.rx(c, iq, want=("ts",))["ts"]
    st = g.RxStream(g.AUTO, g.AUTO, mode, segment_superframes=4)
    st.push(iq[:100 * (c.N + c.cp)])                                  # less than the 160 symbols the detection waits for
    assert st.info().constellation == -1 and len(st.pull()) == 0
    st.push(iq[100 * (c.N + c.cp):])
    st.finish()
    ts = st.pull(); info = st.info(); st.close()
    assert info.auto_configured == 1 and len(ts) == len(want) and (ts == want).all()
    # noise: no TPS word, the look-ahead doubles three times, then the stream refuses
    rng = np.random.RandomState(2)
    noise = (1e-3 * (rng.randn(40 * 160 * 2112) + 1j * rng.randn(40 * 160 * 2112))).astype(np.complex64)
    st = g.RxStream(g.AUTO, g.AUTO, mode)
    with pytest.raises(g.DvbtError):
        for a in range(0, len(noise), 200000):
            st.push(noise[a:a + 200000])
    assert st.info().status & 4
    st.close()


def test_auto_configured_sharded_stream(po):
    """DVBT_AUTO with rank / world: every rank probes the head of the (same) stream by itself, builds its chains for what the TPS word names and replays the head;
    the ranks' chunks ordered by their packet index are the oracle's TS with the true parameters"""
    const, cr, mode = g.QAM64, g.C7_8, g.T8k
    c = po.cfg(const, cr, mode)
    iq = po.stream_slice(c, 9, 21)
    want = po.rx(c, iq, want=("ts",))["ts"]
    world = 2
    ranks = [g.RxStream(g.AUTO, g.AUTO, mode, segment_superframes=1, hierarchy=g.AUTO, rank=r, world=world) for r in range(world)]
    chunks, step = [], 40 * (c.N + c.cp) + 3
    for a in range(0, len(iq), step):
        for st in ranks:
            st.push(iq[a:a + step])
        for r, st in enumerate(ranks):
            chunks += [(fp, r, b) for fp, b in st.pull_chunks()]
    for r, st in enumerate(ranks):
        st.finish()
        chunks += [(fp, r, b) for fp, b in st.pull_chunks()]
    infos = [st.info() for st in ranks]
    for st in ranks:
        st.close()
    assert all(i.auto_configured == 1 and (i.constellation, i.hierarchy, i.code_rate) == (const, g.NH, cr) and i.status & ~2 == 0 for i in infos)
    chunks.sort(key=lambda t: t[0])
    at = chunks[0][0]
    for fp, r, b in chunks:
        assert fp == at, (fp, at, r)
        at += len(b) // 188
    assert {r for _, r, _ in chunks} == {0, 1}
    ts = np.concatenate([b for _, _, b in chunks])
    assert len(ts) == len(want) > 0 and (ts == want).all()
