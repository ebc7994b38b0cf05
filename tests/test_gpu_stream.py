"""dvbt_rx_stream_*: the streaming entry of the C ABI.  Samples pushed in calls of arbitrary size (down to a handful of OFDM symbols, sizes that
do not divide anything), pieces cut, decoded and stitched inside the library; the TS pulled must be, byte for byte, what ONE chain over the whole
stream delivers.  The reference bytes of every case are the ORACLE's (oracle/o_chain.c over the whole stream, po.rx(...)["ts"]); the HIP single chain
(dvbt_rx_segment_run on all samples at once) is required to equal them too, so a stream test never compares the HIP path with itself alone."""
import numpy as np
import pytest

import gr_dvbt_amd as g

pytestmark = pytest.mark.gpu


def whole(po, const, cr, mode, iq, snr_db=30.0):
    """the stream's TS: the oracle's chain over all samples; the HIP single chain must deliver the same bytes"""
    c = po.cfg(const, cr, mode)
    want = po.rx(c, iq, snr_db=snr_db, want=("ts",))["ts"].copy()
    rx = g.Rx(const, cr, mode, max_samples=len(iq), snr_db=snr_db)
    rx.run(iq)
    ts = rx.tap(g.TAP_TS).copy()
    rx.close()
    assert len(ts) == len(want) and (ts == want).all(), "HIP single chain differs from the oracle"
    return want


def streamed(const, cr, mode, iq, seg_sf, call, pull_every=7, **kw):
    st = g.RxStream(const, cr, mode, segment_superframes=seg_sf, **kw)
    out, k = [], 0
    rng = np.random.RandomState(3)
    pos = 0
    while pos < len(iq):
        n = call if isinstance(call, int) else int(rng.randint(call[0], call[1]))
        st.push(iq[pos:pos + n]); pos += n; k += 1
        if k % pull_every == 0:
            out.append(st.pull())
    st.finish()
    out.append(st.pull())
    info = st.info()
    st.close()
    return np.concatenate(out), info


@pytest.mark.parametrize("const,cr,mode,nsf,seg_sf,call", [
    (g.QAM16, g.C1_2, g.T2k, 13, 2, 4 * 2112),            # 2k: pieces of 2 superframes, 4 symbols per call
    (g.QAM16, g.C1_2, g.T2k, 9, 3, (1000, 50000)),         # ragged call sizes
    (g.QAM64, g.C7_8, g.T8k, 9, 2, 64 * 8448),             # 8k QAM64 7/8 (first superframe start at frame 3): 64 symbols per call
    (g.QAM64, g.C7_8, g.T8k, 7, 1, 33 * 8448 + 5),         # one superframe per piece, odd call size
    (g.QPSK, g.C7_8, g.T8k, 8, 2, 16 * 8448),
    (g.QAM64, g.C3_4, g.T2k, 11, 4, 7777),                 # symbols not byte aligned
])
def test_stream_equals_one_chain(po, const, cr, mode, nsf, seg_sf, call):
    c = po.cfg(const, cr, mode)
    iq = po.stream_slice(c, nsf, 9)
    ref = whole(po, const, cr, mode, iq)
    ts, info = streamed(const, cr, mode, iq, seg_sf, call)
    assert info.status & ~2 == 0, info.status
    assert len(ts) == len(ref) > 0, (len(ts), len(ref))
    assert (ts == ref).all()
    assert info.finished and info.ts_bytes_pulled == len(ref) and info.ts_bytes_ready == 0


@pytest.mark.parametrize("extra_symbols", [5, 60, 100, 200, 300])
def test_stream_ends_anywhere(po, extra_symbols):
    """the end of the stream falls at any distance behind the last piece boundary: a tail too short for a piece of its own extends the piece before"""
    const, cr, mode = g.QAM16, g.C1_2, g.T2k
    c = po.cfg(const, cr, mode)
    L = c.N + c.cp
    full = po.stream_slice(c, 12, 4)
    # piece 0 of a 2-superframe plan is 4 superframes + ...: cut the stream so that it ends `extra_symbols` after a later piece's begin
    for base_sf in (7, 8):
        n = po.STREAM_LEAD_IN + (272 * base_sf + extra_symbols) * L
        iq = np.concatenate([full[:n], np.zeros(3 * c.N, np.complex64)])
        ref = whole(po, const, cr, mode, iq)
        ts, info = streamed(const, cr, mode, iq, 2, 10 * L)
        assert len(ts) == len(ref) > 0 and (ts == ref).all(), (base_sf, extra_symbols, len(ts), len(ref))


def test_short_stream_is_one_segment(po):
    const, cr, mode = g.QAM16, g.C1_2, g.T2k
    c = po.cfg(const, cr, mode)
    iq = po.stream_slice(c, 3, 5)
    ref = whole(po, const, cr, mode, iq)
    ts, info = streamed(const, cr, mode, iq, 4, 5000)
    assert len(ts) == len(ref) > 0 and (ts == ref).all()


def test_stream_from_device_memory(po):
    import torch
    const, cr, mode = g.QAM64, g.C7_8, g.T8k
    c = po.cfg(const, cr, mode)
    iq = po.stream_slice(c, 7, 9)
    ref = whole(po, const, cr, mode, iq)
    dev = torch.from_numpy(iq.view(np.float32)).cuda()
    torch.cuda.synchronize()
    st = g.RxStream(const, cr, mode, segment_superframes=2)
    step, out = 64 * 8448, []
    for a in range(0, len(iq), step):
        n = min(step, len(iq) - a)
        st.push_device(dev.data_ptr() + 8 * a, n)
        out.append(st.pull())
    st.finish()
    out.append(st.pull())
    st.close()
    ts = np.concatenate(out)
    assert len(ts) == len(ref) and (ts == ref).all()


def test_noisy_stream_post_rs_equal(po):
    """AWGN at 22 dB on 8k QAM64 7/8: the Viterbi and RS decoders are busy; the pieces' packets must still be the single chain's"""
    const, cr, mode = g.QAM64, g.C7_8, g.T8k
    c = po.cfg(const, cr, mode)
    iq = po.channel(po.stream_slice(c, 6, 9), c.N, snr_db=24.0)
    o = po.rx(c, iq, snr_db=24.0, want=("ts",))
    if len(o["lock_periods"]) != 1:
        pytest.skip("the reference's tracker lost the lock on this noise realisation")
    ref = o["ts"].copy()                                                 # post-RS: the oracle's bytes (the RS decoder removes the float-rounding flips)
    st = g.RxStream(const, cr, mode, segment_superframes=2, snr_db=24.0)
    for a in range(0, len(iq), 50 * 8448):
        st.push(iq[a:a + 50 * 8448])
    st.finish()
    ts = st.pull()
    st.close()
    assert len(ts) == len(ref) and (ts == ref).all()


@pytest.mark.parametrize("const,cr,mode,nsf,seg_sf,world,call", [
    (g.QAM16, g.C1_2, g.T2k, 14, 2, 2, 9 * 2112),
    (g.QAM16, g.C1_2, g.T2k, 11, 1, 3, (1000, 60000)),
    (g.QAM64, g.C7_8, g.T8k, 9, 1, 4, 64 * 8448),
    (g.QAM64, g.C7_8, g.T8k, 10, 2, 2, 40 * 8448 + 3),
])
def test_sharded_stream_equals_one_chain(po, const, cr, mode, nsf, seg_sf, world, call):
    """rank / world in the parameters: every rank is pushed the same stream, decodes the pieces k % world == rank and pulls its packets with
    their index in the stream; the ranks' chunks ordered by that index are the single chain's TS (here: all ranks in one process, on one GPU)"""
    c = po.cfg(const, cr, mode)
    iq = po.stream_slice(c, nsf, 9)
    ref = whole(po, const, cr, mode, iq)
    ranks = [g.RxStream(const, cr, mode, segment_superframes=seg_sf, rank=r, world=world) for r in range(world)]
    chunks = []
    rng = np.random.RandomState(5)
    pos = 0
    while pos < len(iq):
        n = call if isinstance(call, int) else int(rng.randint(call[0], call[1]))
        for st in ranks:
            st.push(iq[pos:pos + n])
        pos += n
        for r, st in enumerate(ranks):
            chunks += [(fp, r, b) for fp, b in st.pull_chunks()]
    for r, st in enumerate(ranks):
        st.finish()
        chunks += [(fp, r, b) for fp, b in st.pull_chunks()]
    infos = [st.info() for st in ranks]
    for st in ranks:
        st.close()
    assert all(i.status & ~2 == 0 for i in infos), [i.status for i in infos]
    chunks.sort(key=lambda t: t[0])
    # contiguous, no packet twice, every rank contributed
    q0 = infos[0].first_ts_packet
    assert chunks[0][0] == q0
    at = q0
    for fp, r, b in chunks:
        assert fp == at, (fp, at, r)
        at += len(b) // 188
    assert len({r for _, r, _ in chunks}) == world
    ts = np.concatenate([b for _, _, b in chunks])
    assert len(ts) == len(ref) > 0 and (ts == ref).all()


def _example():
    from conftest import host_example
    return host_example("rx_stream_example")


@pytest.mark.parametrize("nsf,symbols,out_bytes,seg_sf", [(11, 64, 1 << 22, 4), (11, 4, 188 * 16, 2), (2, 64, 1 << 22, 16), (7, 33, 188 * 700, 1)],
                         ids=["64 symbols per call", "4 symbols per call, 3 KB output buffer (back-pressure)", "stream shorter than the first piece", "one superframe per piece"])
def test_cpp_rx_hip_example(po, tmp_path, nsf, symbols, out_bytes, seg_sf):
    """gr_dvbt_amd/host/rx_stream_example: file_source -> rx_hip -> file_sink driven the way GNU Radio's executor drives a block (forecast, general_work with a
    bounded output buffer, the end of the input signalled the way the shell reads it from the runtime), over the GNU Radio-free mirror of gr::dvbt::rx_hip
    (host/dvbt_blocks.hpp; same logic as host/gr/rx_hip_impl.cc).  The TS FILE -- only what general_work() returned, nothing drained behind the block's
    back -- must be the single chain's, to the last byte: the end of a finite stream is delivered (VERDICT r03 weak 6 / ADVICE r03 item 1), also from a
    stream shorter than the first piece, also with a sink that takes 12 KB per call (the library's FIFO is bounded by back-pressure, not by luck)."""
    import subprocess
    const, cr, mode = g.QAM64, g.C7_8, g.T8k
    c = po.cfg(const, cr, mode)
    iq = po.stream_slice(c, nsf, 6)
    ref = whole(po, const, cr, mode, iq)
    fin, fout = tmp_path / "bb.cf32", tmp_path / "out.ts"
    iq.tofile(fin)
    out = subprocess.check_output([_example(), "8k", "qam64", "7/8", str(fin), str(fout), str(symbols), str(out_bytes), str(seg_sf)], text=True)
    got = np.fromfile(fout, np.uint8)
    assert "status 0" in out and "left inside 0 " in out, out
    assert len(got) == len(ref) > 0 and (got == ref).all(), (len(got), len(ref), out)


def _holed(po, c, nsf, hole_symbol, seed=9):
    iq = po.stream_slice(c, nsf, seed).copy()
    L = c.N + c.cp
    hole = po.STREAM_LEAD_IN + hole_symbol * L
    iq[hole:hole + 30 * L] = 0
    return iq


@pytest.mark.parametrize("hole_symbol,seg_sf", [(272 * 8 + 100, 4), (272 * 12 + 100, 4), (2460, 4), (2460, 2), (272 * 5 + 40, 1), (272 * 9 - 20, 2)],
                         ids=["a piece in the middle", "the final piece", "a superframe start declared on stale counters (96 symbols late)", "the same, pieces of 2 superframes",
                              "stale counters, one superframe per piece", "a hole in front of a superframe start"])
def test_lock_lost_inside_a_piece_is_the_single_chain(po, hole_symbol, seg_sf):
    """a dropout (30 symbols of silence) in the middle of the stream: the reference's tracker loses the lock, searches, locks again; its demodulator hunts the
    superframe start on counters that went on counting through the gap (lib/demod_reference_signals_impl.cc:115-136), the Viterbi decoder is reset there, the
    byte de-interleaver realigned, the descrambler searches again.  The streaming entry leaves its pieces at the piece in which the lock is lost, walks the stream
    window by window with the blocks' state carried along and goes back to pieces once a lock period is established -- the TS is the single chain's (and the
    oracle's), byte for byte.  With the hole 20 symbols behind a superframe start the counters' next "frame 3, symbol 0" lies 96 symbols behind a transmitted
    superframe start: the reference decodes garbage from there to the end of the stream (its descrambler false-locking now and then), and so does the stream --
    its pieces cut on THAT grid (dvbt_rx_cut.start_delay_symbols), the descrambler followed call by call."""
    const, cr, mode = g.QAM16, g.C1_2, g.T2k
    c = po.cfg(const, cr, mode)
    nsf = 15
    iq = _holed(po, c, nsf, hole_symbol)
    L = c.N + c.cp
    ref = whole(po, const, cr, mode, iq)                                   # one chain, every lock period followed; == the oracle's bytes
    ts, info = streamed(const, cr, mode, iq, seg_sf, 64 * L)
    assert info.status & 2, info.status                                 # the loss is reported
    assert len(ts) == len(ref) > 0, (len(ts), len(ref))
    assert (ts == ref).all(), int((ts != ref).sum())
    if hole_symbol not in (2460, 272 * 5 + 40):                        # (those two end in a lock period on a false superframe grid: garbage to the stream's end)
        sent = {bytes(p) for p in po.stream_ts(c, 0, nsf, 9).reshape(-1, 188)}
        good = sum(1 for p in ts.reshape(-1, 188) if bytes(p) in sent)
        assert good >= len(ts) // 188 - 2 * 11 - 16                    # junk only at the junction


@pytest.mark.parametrize("hole_symbol,exact", [(272 * 6 + 100, True), (272 * 10 + 130, True), (272 * 7 + 100, False), (272 * 8 + 100, False)],
                         ids=["back in lock inside rank 1's piece", "inside rank 0's piece", "the descrambler's calls 8 packets off the epoch's grid afterwards", "back in lock at the piece's end"])
def test_lock_lost_in_a_sharded_stream(po, hole_symbol, exact):
    """world 2, the hole in the middle of a piece: the rank that owns the piece walks it to the end of the piece's samples; the reference's chain is back in a lock
    period on the transmitted superframe grid by then, so the ranks' chunks ordered by their packet index are again the single chain's TS.  When the first superframe
    start of the new lock period is the next piece's boundary, the descrambler's re-search straddles the two ranks' samples; when the descrambler finds its NSYNC
    again on the other half of its 16-packet call grid, the stream's last call (and the junction of a later loss) is rounded on a grid the other ranks do not know:
    the rank says so (status bit 5) instead of promising the single chain's bytes there (tools/shard_dbg.py scan: which dropout positions end how)"""
    const, cr, mode = g.QAM16, g.C1_2, g.T2k
    c = po.cfg(const, cr, mode)
    nsf, seg_sf, world = 15, 4, 2
    iq = _holed(po, c, nsf, hole_symbol)
    ref = whole(po, const, cr, mode, iq)
    ranks = [g.RxStream(const, cr, mode, segment_superframes=seg_sf, rank=r, world=world) for r in range(world)]
    chunks = []
    step = 64 * (c.N + c.cp)
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
    assert any(i.status & 2 for i in infos), [i.status for i in infos]
    if not exact:
        assert any(i.status & 32 for i in infos), [i.status for i in infos]
        return
    assert not any(i.status & 32 for i in infos), [i.status for i in infos]
    chunks.sort(key=lambda t: t[0])
    at = None
    for fp, r, b in chunks:                                               # no packet twice: the labels never run backwards into a chunk before
        assert at is None or fp >= at, (fp, at, r)
        at = fp + len(b) // 188
    ts = np.concatenate([b for _, _, b in chunks])
    assert len(ts) == len(ref) > 0 and (ts == ref).all(), (len(ts), len(ref))


@pytest.mark.parametrize("hole", [None, 272 * 5 + 90])
def test_borrowed_device_pushes(po, hole):
    """dvbt_rx_stream_params.borrow_device_pushes: the samples stay where the caller has them (one resident buffer, pushed in 64-symbol calls): pieces that lie in one stretch are
    decoded in place, the walk's windows (the stream's beginning; behind the dropout) and the final piece are gathered from the regions -- the single chain's TS, and the stream says
    how far it has released the caller's memory"""
    import torch
    const, cr, mode = g.QAM64, g.C7_8, g.T8k
    c = po.cfg(const, cr, mode)
    L = c.N + c.cp
    iq = po.stream_slice(c, 9, 9).copy()
    if hole is not None:
        a = po.STREAM_LEAD_IN + hole * L
        iq[a:a + 30 * L] = 0
    ref = whole(po, const, cr, mode, iq)
    dev = torch.from_numpy(iq.view(np.float32)).cuda()
    torch.cuda.synchronize()
    st = g.RxStream(const, cr, mode, segment_superframes=2, borrow=1)
    step, out, rel = 64 * L, [], []
    for a in range(0, len(iq), step):
        n = min(step, len(iq) - a)
        st.push_device(dev.data_ptr() + 8 * a, n)
        out.append(st.pull())
        rel.append(st.info().samples_released)
    st.finish()
    out.append(st.pull())
    st.close()
    ts = np.concatenate(out)
    assert len(ts) == len(ref) > 0 and (ts == ref).all(), (len(ts), len(ref))
    assert all(b >= a for a, b in zip(rel, rel[1:])) and 0 < rel[-1] <= len(iq)      # the release point only moves forward, and it moves
    with pytest.raises(g.DvbtError):                                                   # host pushes are refused on such a stream
        s2 = g.RxStream(const, cr, mode, borrow=1)
        try:
            s2.push(iq[:1000])
        finally:
            s2.close()


@pytest.mark.parametrize("chains,pull_every", [(3, 7), (4, 7), (4, 10000)], ids=["three chains", "four chains", "four chains, pushed without pulls"])
def test_more_chains_per_stream_object(po, chains, pull_every):
    """dvbt_rx_stream_params.chains: three / four chains take the pieces in turn (two / three pieces decode while the next one fills) -- the same bytes: a clean stream of
    one-superframe pieces (every chain busy), and dropouts at several places, where the walk takes over from a lost piece whose samples lie in the buffers of up to three pieces
    behind it (pushed without a single pull: every piece behind the lost one is in flight or filled when the loss is found)"""
    const, cr, mode = g.QAM16, g.C1_2, g.T2k
    c = po.cfg(const, cr, mode)
    L = c.N + c.cp
    iq = po.stream_slice(c, 14, 9)
    ref = whole(po, const, cr, mode, iq)
    ts, info = streamed(const, cr, mode, iq, 1, (1000, 50000), pull_every=pull_every, chains=chains)
    assert info.status & ~2 == 0 and len(ts) == len(ref) > 0 and (ts == ref).all()
    for hole_symbol in (272 * 6 + 100, 272 * 7 + 40, 272 * 9 + 200):
        iq = _holed(po, c, 15, hole_symbol)
        ref = whole(po, const, cr, mode, iq)
        ts, info = streamed(const, cr, mode, iq, 1, 64 * L, pull_every=pull_every, chains=chains)
        assert info.status & 2
        assert len(ts) == len(ref) > 0 and (ts == ref).all(), (hole_symbol, len(ts), len(ref))
    with pytest.raises(RuntimeError):
        g.RxStream(const, cr, mode, chains=5)
