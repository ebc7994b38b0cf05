"""GPU parity of every block of the C ABI (one create/forecast/work/destroy triple per reference
block) against the oracle's restatement of the same block, on seeded inputs, including the
streaming behaviour (state carried across work() calls) and the edge cases the domain has
(erasures/punctured symbols, channel errors up to and beyond the RS correction capability, tags)."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gr_dvbt_amd
    assert gr_dvbt_amd.device_count() > 0
    return gr_dvbt_amd


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _conv_encode(bits):
    """K=7 mother code G1=171, G2=133 (oct); lib/inner_coder_impl.cc:33-48"""
    u = np.concatenate([np.zeros(6, np.uint8), bits.astype(np.uint8)])
    n = len(bits)
    def tap(d): return u[6 - d:6 - d + n]
    x = tap(0) ^ tap(1) ^ tap(2) ^ tap(3) ^ tap(6)
    y = tap(0) ^ tap(2) ^ tap(3) ^ tap(5) ^ tap(6)
    return x, y


def _coded_symbols(po, c, nbytes, ber, seed):
    rng = np.random.RandomState(seed)
    data = rng.randint(0, 256, nbytes).astype(np.uint8)
    bits = np.unpackbits(data)
    x, y = _conv_encode(bits)
    inter = np.empty(2 * len(bits), np.uint8)
    inter[0::2] = x
    inter[1::2] = y
    ln = C.c_int()
    po.lib().o_vit_puncture.restype = C.POINTER(C.c_ubyte)
    pp = po.lib().o_vit_puncture(c.code_rate, C.byref(ln))
    punct = np.array([pp[i] for i in range(ln.value)], np.uint8)
    mask = np.tile(punct, len(inter) // len(punct) + 1)[:len(inter)].astype(bool)
    kept = inter[mask]
    kept = kept ^ (rng.rand(len(kept)) < ber)
    nsym = len(kept) // c.m
    sym = np.zeros(nsym, np.uint8)
    for j in range(c.m):
        sym |= (kept[j:nsym * c.m:c.m].astype(np.uint8) << (c.m - 1 - j))
    return data, sym


def test_fft_block(po, g):
    rng = np.random.RandomState(1)
    for N in (2048, 8192):
        x = (rng.randn(5, N) + 1j * rng.randn(5, N)).astype(np.complex64)
        ref = np.zeros_like(x)
        for i in range(5):
            po.lib().o_fft_forward_shift(N, _p(x[i]), _p(ref[i]))
        b = g.Block("fft", N, 1, 1)
        out = np.zeros_like(x)
        r, cons, _ = b.work(5, 5, x, out)
        assert r == 5 and cons == 5
        assert np.abs(out - ref).max() <= 1e-5 * np.abs(ref).max()
        assert np.abs(out - np.fft.fftshift(np.fft.fft(x.astype(np.complex128), axis=1), axes=1)).max() <= 1e-5 * np.abs(ref).max()
        b.close()
    with pytest.raises(g.DvbtError):
        g.Block("fft", 1000, 1, 1)


@pytest.mark.parametrize("const,mode", [(0, 1), (1, 0), (2, 1)])
def test_demap_block_bit_exact(po, g, const, mode):
    c = po.cfg(const, po.C1_2, mode)
    pts = np.zeros(c.csize, np.complex64)
    po.lib().o_constellation(C.byref(c), C.c_float(1.0), _p(pts))
    rng = np.random.RandomState(2)
    n = 3
    lab = rng.randint(0, c.csize, (n, c.payload))
    x = (pts[lab] + 0.35 * c.norm * (rng.randn(n, c.payload) + 1j * rng.randn(n, c.payload))).astype(np.complex64)
    x[0, :c.csize] = pts                                  # exact points
    x[0, c.csize:2 * c.csize] = (pts + np.roll(pts, 1)) / 2     # exact midpoints: the first-minimum tie rule decides
    ref = np.zeros((n, c.payload), np.uint8)
    po.lib().o_demap(C.byref(c), _p(pts), _p(x), _p(ref), C.c_size_t(n * c.payload))
    b = g.Block("demap", c.payload, const, 0, mode, 1.0)
    out = np.zeros_like(ref)
    assert b.work(n, n, x, out)[0] == n
    assert (out == ref).all()
    b.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_symbol_deinterleaver_block(po, g, mode):
    c = po.cfg(po.QAM16, po.C1_2, mode)
    H = np.zeros(c.payload, np.int32)
    po.lib().o_sym_H(C.byref(c), _p(H))
    rng = np.random.RandomState(3)
    idx = [0, 1, 2, 67, 33, 10]
    x = rng.randint(0, 64, (len(idx), c.payload)).astype(np.uint8)
    ref = np.zeros_like(x)
    for i, si in enumerate(idx):
        po.lib().o_sym_interleave(C.byref(c), _p(H), _p(x[i]), _p(ref[i]), si, 0)
    b = g.Block("symbol_inner_interleaver", c.payload, mode, 0)
    out = np.zeros_like(x)
    r, _, _ = b.work(len(idx), len(idx), x, out, tags=[(i, g.TAG_SYMBOL_INDEX, si) for i, si in enumerate(idx)])
    assert r == len(idx) and (out == ref).all()
    with pytest.raises(g.DvbtError):
        b.work(2, 2, x, out, tags=[(0, g.TAG_SYMBOL_INDEX, 0)])      # a tag per item is mandatory
    b.close()
    # direction 1 followed by direction 0 is the identity (TX counts symbols itself)
    tx = g.Block("symbol_inner_interleaver", c.payload, mode, 1)
    mid = np.zeros_like(x)
    tx.work(len(idx), len(idx), x, mid)
    rxb = g.Block("symbol_inner_interleaver", c.payload, mode, 0)
    back = np.zeros_like(x)
    rxb.work(len(idx), len(idx), mid, back, tags=[(i, g.TAG_SYMBOL_INDEX, i) for i in range(len(idx))])
    assert (back == x).all()
    tx.close(); rxb.close()


@pytest.mark.parametrize("const", [0, 1, 2])
def test_bit_deinterleaver_block(po, g, const):
    c = po.cfg(const, po.C1_2, po.T2k)
    rng = np.random.RandomState(4)
    x = rng.randint(0, c.csize, (4, c.payload)).astype(np.uint8)
    ref = np.zeros_like(x)
    po.lib().o_bit_deinterleave(C.byref(c), _p(x), _p(ref), C.c_size_t(x.size))
    b = g.Block("bit_inner_deinterleaver", c.payload, const, 0, po.T2k)
    out = np.zeros_like(x)
    assert b.work(4, 4, x, out)[0] == 4 and (out == ref).all()
    # inverse of the TX interleaver
    fw = np.zeros_like(x)
    po.lib().o_bit_interleave(C.byref(c), _p(x), _p(fw), C.c_size_t(x.size))
    b.work(4, 4, fw, out)
    assert (out == x).all()
    b.close()


@pytest.mark.parametrize("const,cr,ber", [(1, 0, 0.0), (1, 0, 0.04), (2, 4, 0.0), (2, 4, 0.004), (0, 2, 0.02), (2, 3, 0.01), (1, 1, 0.03)])
def test_viterbi_block_streaming_bit_exact(po, g, const, cr, ber):
    """Chunk-parallel GPU decode == the streaming reference algorithm (oracle pinned to the reference's
    own SSE2 kernels), with channel errors, fed through several work() calls of different sizes."""
    c = po.cfg(const, cr, po.T2k)
    bsize = 768
    d_nsym, d_nout = bsize * c.n // c.m, bsize * c.k // 8
    nblocks = 23
    data, sym = _coded_symbols(po, c, d_nout * nblocks + 64, ber, 10 + cr)
    sym = sym[:d_nsym * nblocks].copy()
    ref = np.zeros(d_nout * nblocks + 64, np.uint8)
    n_ref = po.lib().o_viterbi_decode(C.byref(c), bsize, _p(sym), len(sym), _p(ref))
    ntb = g.get_dims(const, cr, po.T2k).ntraceback
    assert n_ref == d_nout * nblocks - ntb
    if ber == 0.0:
        assert (ref[:n_ref] == data[:n_ref]).all()
    b = g.Block("viterbi_decoder", const, 0, cr, bsize, 0, -1)
    assert b.forecast(d_nout) == d_nsym
    outs, pos, first = [], 0, True
    for nb in (1, 2, 7, 1, 12):
        o = np.zeros(nb * d_nout, np.uint8)
        tags = [(0, g.TAG_SUPERFRAME_START, 0xaa)] if first else []
        r, cons, tout = b.work(nb * d_nout, nb * d_nsym, sym[pos:pos + nb * d_nsym], o, tags=tags)
        assert cons == nb * d_nsym and r == nb * d_nout - (ntb if first else 0)
        assert (tout == [(0, g.TAG_SUPERFRAME_START, 1)]) == first
        outs.append(o[:r]); pos += cons; first = False
    out = np.concatenate(outs)
    assert len(out) == n_ref and (out == ref[:n_ref]).all()
    # a superframe_start tag in the middle of the window: consume up to it, produce nothing, then restart
    o = np.zeros(2 * d_nout, np.uint8)
    r, cons, _ = b.work(2 * d_nout, 2 * d_nsym, sym[:2 * d_nsym], o, tags=[(100, g.TAG_SUPERFRAME_START, 0xaa)])
    assert r == 0 and cons == 100
    r, cons, _ = b.work(2 * d_nout, 2 * d_nsym, sym[:2 * d_nsym], o, tags=[(0, g.TAG_SUPERFRAME_START, 0xaa)])
    assert r == 2 * d_nout - ntb and (o[:r] == ref[:r]).all()
    b.close()


def test_viterbi_block_garbage_input(po, g):
    """Uniformly random input symbols (no code structure at all): worst case for survivor merging and
    tie-breaking; the chunked decode reproduces the streaming decode on this input.  (A dozen chunk starts: on garbage
    about one start in two hundred does NOT merge inside the default warm-up of 72 windows -- DESIGN.md 2,
    tests/test_gpu_warmup.py and tests/test_viterbi_warmup_model.py hold the statistics and dvbt_rx_params.viterbi_warm_windows.)"""
    c = po.cfg(po.QAM64, po.C7_8, po.T8k)
    rng = np.random.RandomState(9)
    sym = rng.randint(0, 64, 1024 * 12).astype(np.uint8)
    ref = np.zeros(672 * 12, np.uint8)
    n_ref = po.lib().o_viterbi_decode(C.byref(c), 768, _p(sym), len(sym), _p(ref))
    b = g.Block("viterbi_decoder", 2, 0, 4, 768, 0, -1)
    out = np.zeros(672 * 12, np.uint8)
    r, _, _ = b.work(672 * 12, len(sym), sym, out, tags=[(0, g.TAG_SUPERFRAME_START, 0xaa)])
    assert r == n_ref and (out[:r] == ref[:r]).all()
    b.close()


def test_convolutional_deinterleaver_block(po, g):
    rng = np.random.RandomState(6)
    x = rng.randint(0, 256, 1632 * 14).astype(np.uint8)
    ref = np.zeros_like(x)
    po.lib().o_conv_deinterleave(_p(x), _p(ref), C.c_size_t(len(x)))
    b = g.Block("convolutional_deinterleaver", 136, 12, 17)
    assert b.forecast(2) == 2 * 1632
    outs, pos = [], 0
    for items in (2, 4, 2, 6):
        o = np.zeros(items * 1632, np.uint8)
        r, cons, _ = b.work(items, items * 1632, x[pos:pos + items * 1632], o)
        assert r == items and cons == items * 1632
        outs.append(o); pos += cons
    assert (np.concatenate(outs) == ref).all()
    assert b.work(3, 3 * 1632, x, np.zeros(3 * 1632, np.uint8))[0] == 2        # output multiple of 2
    o = np.zeros(2 * 1632, np.uint8)
    assert b.work(2, 2 * 1632, x, o, tags=[(77, g.TAG_SUPERFRAME_START, 1)])[:2] == (0, 77)
    b.close()
    with pytest.raises(g.DvbtError):
        g.Block("convolutional_deinterleaver", 100, 12, 17)


@pytest.mark.parametrize("compat", [0, 1])
def test_reed_solomon_block(po, g, compat):
    L = po.lib()
    rs = po.RS()
    L.o_rs_init(C.byref(rs))
    rng = np.random.RandomState(7 + compat)
    items = 12
    words = np.zeros((items * 8, 204), np.uint8)
    for w in range(items * 8):
        cw = np.zeros(255, np.uint8)
        cw[51:239] = rng.randint(0, 256, 188)
        par = np.zeros(16, np.uint8)
        L.o_rs_encode(C.byref(rs), _p(cw), _p(par))
        cw[239:] = par
        nerr = [0, 0, 1, 2, 5, 8, 8, 9, 10, 16, 3, 7][w % 12]
        for pos in rng.choice(np.arange(51, 255), nerr, replace=False):
            cw[pos] ^= rng.randint(1, 256)
        words[w] = cw[51:]
    words[5] = rng.randint(0, 256, 204)                  # pure garbage word
    ref = np.zeros((items * 8, 188), np.uint8)
    nf, nc = C.c_int(), C.c_int()
    L.o_rs_dec_block(C.byref(rs), _p(words), _p(ref), C.c_size_t(items * 8), compat, C.byref(nf), C.byref(nc))
    assert nf.value > 0 and nc.value > 0
    b = g.Block("reed_solomon_dec", 2, 8, 0x11d, 255, 239, 8, 51, 8, compat)
    out = np.zeros_like(ref)
    r, cons, _ = b.work(items, items, words, out)
    assert r == items and cons == items and (out == ref).all()
    b.close()
    with pytest.raises(g.DvbtError):
        g.Block("reed_solomon_dec", 2, 8, 0x11d, 255, 223, 16, 0, 8, 0)


def test_energy_descramble_block(po, g):
    ts = po.make_ts(8 * 20, 3)
    disp = np.zeros_like(ts)
    po.lib().o_energy_dispersal(_p(ts), _p(disp), C.c_size_t(len(ts) // 188))
    x = np.concatenate([disp[188 * 3:], disp[:188 * 3]])          # group start not at item start
    nitems = len(x) // 1504
    ref = np.zeros(len(x), np.uint8)
    n_ref = po.lib().o_energy_descramble(_p(x), C.c_size_t(nitems), _p(ref))
    b = g.Block("energy_descramble", 8)
    out = np.zeros(len(x), np.uint8)
    nout = (nitems // 4) * 4 * 1504
    r, cons, _ = b.work(nout, nitems, x, out)
    assert cons == nout // 1504 - 2 and r == cons * 1504
    assert (out[:r] == ref[:r]).all() and r <= n_ref
    assert (out[:r].reshape(-1, 188)[:, 0] == 0x47).all()
    b.close()


def test_acquisition_block_streaming(po, g):
    """A1 alone, fed in work() calls of arbitrary size: items, consumption and the sync_start tag."""
    c = po.cfg(po.QAM16, po.C1_2, po.T2k)
    iq = po.tx(c, po.make_ts(504, 8), lead_in=700, tail=3 * c.N)[:120 * (c.N + c.cp)]
    o = po.rx(c, iq, want=("acq",))
    b = g.Block("ofdm_sym_acquisition", 1, c.N, c.Kmax + 1, c.cp, 30.0)
    pos, items, first = 0, [], True
    for want in (1, 3, 10, 40, 200):
        need = b.forecast(want)
        chunk = iq[pos:pos + need]
        out = np.zeros((want, c.N), np.complex64)
        r, cons, tags = b.work(want, len(chunk), chunk, out)
        if r == 0 and cons == 0:
            break
        assert ((0, g.TAG_SYNC_START, 1) in tags) == first
        first = False
        items.append(out[:r]); pos += cons
    got = np.concatenate(items)
    n = min(len(got), len(o["acq"]))
    assert n >= 100
    assert np.abs(got[:n] - o["acq"][:n]).max() <= 1e-6 * np.abs(o["acq"]).max()
    b.close()


def test_demod_block_streaming(po, g):
    """A3 alone over FFT items (taken from the oracle so only this block is under test): equalised
    carriers within tolerance, the superframe_start tag on the right item, symbol_index per item."""
    c = po.cfg(po.QAM16, po.C1_2, po.T2k)
    iq = po.tx(c, po.make_ts(504 * 2 + 100, 9), lead_in=900, tail=3 * c.N)
    o = po.rx(c, iq, want=("fft", "eq"))
    fft, fo = o["fft"], o["first_out_symbol"]
    b = g.Block("demod_reference_signals", 8, c.N, c.payload, po.QAM16, 0, po.C1_2, po.C1_2, 0, po.T2k, 0, 0)
    pos, outs, all_tags, produced = 0, [], [], 0
    for want in (1, 5, 100, 200, 300):
        avail = min(want + 1, len(fft) - pos)
        if avail < 2:
            break
        out = np.zeros((want, c.payload), np.complex64)
        r, cons, tags = b.work(want, avail, fft[pos:pos + avail], out, tags=[(0, g.TAG_SYNC_START, 1)] if pos == 0 else [])
        assert cons == avail - 1
        all_tags += [(off + produced, k, v) for off, k, v in tags]
        outs.append(out[:r]); produced += r; pos += cons
    got = np.concatenate(outs)
    n = min(len(got), len(o["eq"]))
    assert n > 200
    d = got[:n] - o["eq"][:n]
    assert max(np.abs(d.real).max(), np.abs(d.imag).max()) <= 1e-3 * 2 * c.norm
    sf = [t for t in all_tags if t[1] == g.TAG_SUPERFRAME_START]
    assert sf == [(0, g.TAG_SUPERFRAME_START, 0xaa)]
    si = [t[2] for t in all_tags if t[1] == g.TAG_SYMBOL_INDEX]
    assert si[:n] == list(o["sym_index"][fo:fo + n])
    b.close()


def test_cpp_host_mirror_example(po, g, tmp_path):
    """gr_dvbt_amd/host/rx_flowgraph_example: the back half of apps/dvbt_rx_demo.grc written against the C++ mirror of the
    gr::dvbt block classes (same make() calls and general_work contract), fed with the oracle's bit-de-interleaver output."""
    import os
    import subprocess
    from conftest import host_example
    exe = host_example("rx_flowgraph_example")
    c = po.cfg(po.QAM16, po.C1_2, po.T2k)
    ts = po.make_ts(504 * 3, 9)
    iq = po.tx(c, ts, lead_in=500, tail=3 * c.N)
    o = po.rx(c, iq, want=("bitdeint", "ts"))
    fin, fout = tmp_path / "bitdeint.bin", tmp_path / "out.ts"
    o["bitdeint"].tofile(fin)
    subprocess.check_call([exe, str(fin), str(fout)], stdout=subprocess.DEVNULL)
    got = np.fromfile(fout, np.uint8)
    n = min(len(got), len(o["ts"]))
    assert n > 100000 and abs(len(got) - len(o["ts"])) <= 4 * 1504
    assert (got[:n] == o["ts"][:n]).all()


def test_wave_peak_detector(g):
    """the sequential trackers' wavefront-wide peak detector (k_frontend.hpp::peak_detect16_wave: IIR average as the one recurrence, threshold tests as ballots,
    the state machine on 16-bit masks with DPP prefix maxima) against the sample-by-sample restatement of peak_detect_process
    (lib/ofdm_sym_acquisition_impl.cc:72-146) on 60,000 windows: noise around a carried average of either sign, triangles like a CP peak, plateaus and
    exact ties (quantised values), windows that end inside an open peak, several peaks per window, the -3e38 sentinel of lags in front of the stream."""
    rng = np.random.RandomState(11)
    n = 60000
    lam = np.empty((n, 16), np.float32)
    avg = np.empty(n, np.float32)
    kind = rng.randint(0, 6, n)
    for k in range(n):
        t = kind[k]
        base = rng.uniform(-300, 50)
        if t == 0:
            x = base + rng.randn(16) * rng.choice([0.1, 3, 30])
        elif t == 1:                                                     # a triangle (the CP peak) on a noisy floor
            c = rng.uniform(-4, 20); h = rng.uniform(10, 400)
            x = base + np.maximum(0, h * (1 - np.abs(np.arange(16) - c) / rng.uniform(2, 12))) + rng.randn(16) * rng.choice([0.5, 5, 20])
        elif t == 2:                                                     # quantised: plateaus and exact ties
            x = np.round((base + rng.randn(16) * 20) / 16) * 16
        elif t == 3:                                                     # positive values, several peaks
            x = np.abs(rng.randn(16)) * rng.choice([1, 50]) * (rng.rand(16) > 0.4)
        elif t == 4:                                                     # lags in front of the stream's first sample
            x = base + rng.randn(16) * 10; x[:rng.randint(1, 16)] = -3.0e38
        else:                                                            # ramps: a window that ends inside an open peak
            x = base + np.arange(16) * rng.uniform(-30, 30) + rng.randn(16) * 3
        lam[k] = x.astype(np.float32)
        avg[k] = np.float32(rng.choice([0.0, base, base * 0.5, -base, x[0]]) + rng.randn() * 5)
    out = np.zeros((n, 4), np.int32)
    avg_out = np.zeros((n, 2), np.float32)
    L = g.lib()
    L.dvbt_debug_peak_detect.restype = C.c_int
    L.dvbt_debug_peak_detect.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    assert L.dvbt_debug_peak_detect(lam.ctypes.data, avg.ctypes.data, n, out.ctypes.data, avg_out.ctypes.data) == 0
    assert (out[:, 0] == out[:, 2]).all(), np.flatnonzero(out[:, 0] != out[:, 2])[:5]
    assert (out[:, 1] == out[:, 3]).all(), np.flatnonzero(out[:, 1] != out[:, 3])[:5]
    assert (avg_out[:, 0].view(np.uint32) == avg_out[:, 1].view(np.uint32)).all()
    # the cases are not degenerate: windows without a peak, with one, with several
    assert (out[:, 0] == 0).sum() > 1000 and (out[:, 0] == 1).sum() > 1000 and (out[:, 0] >= 2).sum() > 1000


# ---------------------------------------------------------------- hierarchical modes (rows A0 / A4 / A6: dvbt_config.cc:213-225, dvbt_demap_impl.cc:117-165,
# bit_inner_deinterleaver_impl.cc:148-184)
@pytest.mark.parametrize("const,hier,mode", [(1, 2, 0), (1, 3, 1), (2, 2, 1), (2, 3, 0), (2, 1, 0)])
def test_demap_block_hierarchical_constellations(po, g, const, hier, mode):
    """alpha = 2 / 4: the constellation's two halves per axis lie (alpha - 1) units further apart.  Random carriers, exact points, exact midpoints (the first
    strict minimum decides), carriers inside the centre gap, far outside the grid, and a dense sweep across every decision boundary of one axis."""
    c = po.cfg(const, po.C1_2, mode, hierarchy=hier)
    pts = np.zeros(c.csize, np.complex64)
    po.lib().o_constellation(C.byref(c), C.c_float(1.0), _p(pts))
    rng = np.random.RandomState(12)
    n = 4
    lab = rng.randint(0, c.csize, (n, c.payload))
    x = (pts[lab] + 0.45 * c.norm * (rng.randn(n, c.payload) + 1j * rng.randn(n, c.payload))).astype(np.complex64)
    x[0, :c.csize] = pts
    x[0, c.csize:2 * c.csize] = (pts + np.roll(pts, 1)) / 2
    x[0, 2 * c.csize:3 * c.csize] = (pts + np.roll(pts, 3)) / 2
    x[1, :400] = ((rng.rand(400) - 0.5) + 1j * (rng.rand(400) - 0.5)).astype(np.complex64) * 2 * c.alpha * c.norm      # inside the centre gap
    x[1, 400:800] = ((rng.randn(400) + 1j * rng.randn(400)) * 40).astype(np.complex64)                                 # far outside
    sweep = np.linspace(-12 * c.norm, 12 * c.norm, 600).astype(np.float32)
    x[2, :600] = sweep + 1j * np.float32(0.3 * c.norm)
    x[2, 600:1200] = np.float32(-1.7 * c.norm) + 1j * sweep
    ref = np.zeros((n, c.payload), np.uint8)
    po.lib().o_demap(C.byref(c), _p(pts), _p(x), _p(ref), C.c_size_t(n * c.payload))
    b = g.Block("demap", c.payload, const, hier, mode, 1.0)
    out = np.zeros_like(ref)
    assert b.work(n, n, x, out)[0] == n
    assert (out == ref).all(), np.argwhere(out != ref)[:5]
    b.close()


@pytest.mark.parametrize("const,hier", [(1, 1), (1, 2), (2, 2), (2, 3)])
def test_bit_deinterleaver_block_hierarchical_two_outputs(po, g, const, hier):
    c = po.cfg(const, po.C1_2, po.T2k, hierarchy=hier)
    rng = np.random.RandomState(13)
    x = rng.randint(0, c.csize, (5, c.payload)).astype(np.uint8)
    rh, rl = np.zeros_like(x), np.zeros_like(x)
    po.lib().o_bit_deinterleave_hier(C.byref(c), _p(x), _p(rh), _p(rl), C.c_size_t(x.size))
    b = g.Block("bit_inner_deinterleaver", c.payload, const, hier, po.T2k)
    L = g.lib()
    L.dvbt_bit_inner_deinterleaver_work_hier.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    oh, ol = np.zeros_like(x), np.full_like(x, 0xee)
    assert L.dvbt_bit_inner_deinterleaver_work_hier(b.h, 5, 5, _p(x), _p(oh), _p(ol), None) == 5
    assert (oh == rh).all() and (ol == rl).all()
    assert oh.max() == 3 and (ol.max() == 3 if const == 2 else ol.max() == 0)
    # the one-output entry delivers port 0, as a flowgraph that connects only the first output gets it
    o0 = np.zeros_like(x)
    assert b.work(5, 5, x, o0)[0] == 5 and (o0 == rh).all()
    b.close()
    # a non-hierarchical block has one output; hierarchical QPSK does not exist (the reference divides by zero)
    nh = g.Block("bit_inner_deinterleaver", c.payload, const, 0, po.T2k)
    assert L.dvbt_bit_inner_deinterleaver_work_hier(nh.h, 5, 5, _p(x), _p(oh), _p(ol), None) < 0
    nh.close()
    with pytest.raises(g.DvbtError):
        g.Block("bit_inner_deinterleaver", c.payload, 0, hier, po.T2k)
