"""ctypes binding of oracle/liboracle.so (the CPU restatement of the reference RX path).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never from the product package (dvbt_amd/).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libdviterbi_ref.so")

QPSK, QAM16, QAM64 = 0, 1, 2
NH = 0
C1_2, C2_3, C3_4, C5_6, C7_8 = 0, 1, 2, 3, 4
T2k, T8k = 0, 1
G1_32, G1_16, G1_8, G1_4 = 0, 1, 2, 3


def build(force=False):
    if force or not os.path.exists(_LIB) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB)
            for f in os.listdir(_HERE) if f.endswith((".c", ".h"))):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/lib") and (force or not os.path.exists(_REF)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


class Cfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "constellation", "hierarchy", "code_rate", "guard", "mode", "include_cell_id", "cell_id",
        "N", "cp", "Kmin", "Kmax", "payload", "zeros_left", "zeros_right",
        "csize", "step", "m", "alpha", "k", "n")] + [
        ("norm", C.c_float), ("n_cpilot", C.c_int), ("n_tps", C.c_int), ("n_spilot", C.c_int),
        ("cpilot", C.POINTER(C.c_int)), ("tps", C.POINTER(C.c_int))]


class RS(C.Structure):
    _fields_ = [("exp", C.c_ubyte * 256), ("log", C.c_ubyte * 256), ("l", C.c_ubyte * 256),
                ("g", C.c_ubyte * 17)]


class VitCore(C.Structure):
    _fields_ = [("metric", C.c_ubyte * 64), ("path", C.c_ubyte * 64), ("pp", (C.c_ubyte * 64) * 24),
                ("store_pos", C.c_int), ("ntraceback", C.c_int)]


class Taps(C.Structure):
    _fields_ = [
        ("acq_out", C.c_void_p), ("acq_cap", C.c_size_t), ("acq_n", C.c_size_t),
        ("fft_out", C.c_void_p), ("fft_cap", C.c_size_t), ("fft_n", C.c_size_t),
        ("eq_out", C.c_void_p), ("eq_cap", C.c_size_t), ("eq_n", C.c_size_t),
        ("demap_out", C.c_void_p), ("symdeint_out", C.c_void_p), ("bitdeint_out", C.c_void_p),
        ("sym_cap", C.c_size_t), ("sym_n", C.c_size_t),
        ("vit_out", C.c_void_p), ("vit_cap", C.c_size_t), ("vit_n", C.c_size_t),
        ("deint_out", C.c_void_p), ("deint_cap", C.c_size_t), ("deint_n", C.c_size_t),
        ("rs_out", C.c_void_p), ("rs_cap", C.c_size_t), ("rs_n", C.c_size_t),
        ("ts_out", C.c_void_p), ("ts_cap", C.c_size_t), ("ts_n", C.c_size_t),
        ("cp_start", C.c_void_p), ("epsilon", C.c_void_p), ("sym_index", C.c_void_p),
        ("meta_cap", C.c_size_t),
        ("first_out_symbol", C.c_int), ("n_acquired", C.c_int),
        ("rs_fail", C.c_int), ("rs_corr", C.c_int),
        ("t_stage", C.c_double * 10), ("ts_first_packet", C.c_longlong), ("stream_rs_items", C.c_longlong),
        ("freq_offset", C.c_void_p), ("call_pos", C.c_void_p), ("sync_flag", C.c_void_p), ("bitdeint_lp_out", C.c_void_p), ("sf_flag", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        L = _lib
        L.o_tx_symbols_for_packets.restype = C.c_size_t
        L.o_tx_symbols_for_packets.argtypes = [C.POINTER(Cfg), C.c_size_t]
        L.o_tx_generate.restype = C.c_size_t
        L.o_tx_generate.argtypes = [C.POINTER(Cfg), C.c_void_p, C.c_size_t, C.c_float, C.c_void_p,
                                    C.c_size_t, C.c_void_p]
        L.o_viterbi_decode.restype = C.c_size_t
        L.o_viterbi_decode.argtypes = [C.POINTER(Cfg), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.o_energy_descramble.restype = C.c_size_t
        L.o_energy_descramble.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.o_rx_run.restype = C.c_int
        L.o_rx_run.argtypes = [C.POINTER(Cfg), C.c_void_p, C.c_size_t, C.c_float, C.c_int, C.c_int,
                               C.POINTER(Taps)]
        L.o_rx_run_cut.restype = C.c_int
        L.o_rx_run_cut.argtypes = [C.POINTER(Cfg), C.c_void_p, C.c_size_t, C.c_float, C.c_int, C.c_int, C.c_longlong,
                                   C.POINTER(Taps)]
        L.o_tx_generate_from.restype = C.c_size_t
        L.o_tx_generate_from.argtypes = [C.POINTER(Cfg), C.c_void_p, C.c_size_t, C.c_size_t, C.c_float, C.c_void_p,
                                         C.c_size_t, C.c_void_p]
        L.o_rs_decode.restype = C.c_int
        L.o_vit_get_output.restype = C.c_ubyte
        L.o_resample_scale.restype = C.c_size_t
        L.o_resample_scale.argtypes = [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.o_resampler_nout.restype = C.c_size_t
        L.o_resampler_nout.argtypes = [C.c_int, C.c_int, C.c_size_t]
        L.o_acq_new.restype = C.c_void_p
        L.o_demod_new.restype = C.c_void_p
    return _lib


def ref_lib():
    """The reference's own d_viterbi.c/d_tab.c compiled unmodified (oracle/_ref)."""
    build()
    if not os.path.exists(_REF):
        return None
    return C.CDLL(_REF)


def cfg(constellation, code_rate, mode, guard=G1_32, hierarchy=NH, include_cell_id=0, cell_id=0):
    c = Cfg()
    lib().o_cfg_init(C.byref(c), constellation, hierarchy, code_rate, guard, mode, include_cell_id, cell_id)
    return c


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def make_ts(npackets, seed):
    """188-byte packets, byte0 = 0x47, payload uniform from a seeded generator."""
    rng = np.random.RandomState(seed)
    ts = rng.randint(0, 256, size=(npackets, 188)).astype(np.uint8)
    ts[:, 0] = 0x47
    return ts.reshape(-1)


def packets_per_superframe(c):
    return 272 * (c.payload * c.m * c.k // c.n) // (204 * 8)


def stream_ts(c, first_sf, n_sf, seed):
    """TS of superframes [first_sf, first_sf + n_sf) of THE synthetic stream `seed`: every superframe's packets come
    from their own generator, so any part of a long stream can be produced without the rest (bench.py: every rank
    generates only the piece it decodes)."""
    pps = packets_per_superframe(c)
    out = np.empty((n_sf * pps, 188), np.uint8)
    for j in range(n_sf):
        rng = np.random.RandomState((seed * 1000003 + first_sf + j) % (2 ** 31))
        out[j * pps:(j + 1) * pps] = rng.randint(0, 256, size=(pps, 188)).astype(np.uint8)
    out[:, 0] = 0x47
    return out.reshape(-1)


STREAM_LEAD_IN = 1000


def stream_len(c, n_sf, lead_in=STREAM_LEAD_IN):
    return lead_in + n_sf * 272 * (c.N + c.cp) + 3 * c.N


def stream_slice(c, n_sf, seed, begin=0, end=None, lead_in=STREAM_LEAD_IN):
    """Samples [begin, end) of the synthetic stream (lead_in zeros, n_sf superframes of stream_ts, 3N zeros).
    A slice that starts inside superframe j > 0 is generated from the start of superframe j (or j - 1 when it begins
    within the first 16 symbols of j): the generator's interleaver and encoder start from zero there, which changes
    only the first 11 RS words' worth of symbols of that superframe -- ahead of the slice or, for a cut piece, deep in
    its pre-roll's discarded output."""
    L = c.N + c.cp
    total = stream_len(c, n_sf, lead_in)
    end = total if end is None else min(end, total)
    out = np.zeros(end - begin, np.complex64)
    sym_b = max(0, (begin - lead_in) // L)
    j0 = sym_b // 272
    if j0 > 0 and sym_b - 272 * j0 < 16:
        j0 -= 1
    j1 = min(n_sf, -(-max(end - lead_in, 0) // (272 * L)))
    if j1 > j0:
        body = tx(c, stream_ts(c, j0, j1 - j0, seed), packet0=j0 * packets_per_superframe(c))
        s0 = lead_in + j0 * 272 * L                                  # stream sample of body[0]
        a, b = max(begin, s0), min(end, s0 + len(body))
        if b > a:
            out[a - begin:b - begin] = body[a - s0:b - s0]
    return out


def clock_offset(iq, ppm, taps=16):
    """The stream as a receiver whose sample clock is off by `ppm` sees it: y[n] = x(n (1 + ppm 1e-6)), windowed-sinc interpolation
    (test utility: a slowly sliding symbol timing for the drift tests)."""
    iq = np.ascontiguousarray(iq, dtype=np.complex64)
    n_out = int((len(iq) - taps - 2) / (1.0 + ppm * 1e-6))
    out = np.empty(n_out, np.complex64)
    k = np.arange(-(taps // 2 - 1), taps // 2 + 1)                  # taps around the sampling instant
    step = 1 << 20
    for a in range(0, n_out, step):
        b = min(n_out, a + step)
        t = np.arange(a, b, dtype=np.float64) * (1.0 + ppm * 1e-6) + (taps // 2)
        base = np.floor(t).astype(np.int64)
        frac = (t - base)[:, None]
        x = k[None, :] - frac
        h = np.sinc(x) * (0.54 + 0.46 * np.cos(np.pi * x / (taps // 2 + 1)))   # Hamming-windowed sinc
        idx = base[:, None] + k[None, :]
        out[a:b] = (iq[idx] * h.astype(np.float32)).sum(axis=1)
    return out


def channel(iq, N, cfo=0.0, echoes=(), snr_db=None, seed=5, ref_span=(1000, 101000)):
    """Test utility: what a real capture does to the loopback baseband.  echoes: (delay in samples, complex amplitude) of static
    extra paths; cfo: carrier offset in subcarrier spacings (fractional part -> ofdm_sym_acquisition's epsilon, integer part ->
    pilot_gen::process_cpilot_data); AWGN at snr_db relative to the power of iq[ref_span] after the channel.  Computed in
    double, returned as complex64."""
    x = np.asarray(iq).astype(np.complex128)
    if echoes:
        y = x.copy()
        for d, a in echoes:
            y[d:] += a * x[:-d]
        x = y
    if cfo:
        x *= np.exp(2j * np.pi * cfo / N * np.arange(len(x), dtype=np.float64))
    if snr_db is not None:
        rng = np.random.RandomState(seed)
        p = np.mean(np.abs(x[ref_span[0]:ref_span[1]]) ** 2)
        sig = np.sqrt(p / (10 ** (snr_db / 10)) / 2)
        x = x + sig * (rng.randn(len(x)) + 1j * rng.randn(len(x)))
    return x.astype(np.complex64)


def _cos_sin_exact(theta):
    """cos, sin of a small angle by their Taylor series in plain double arithmetic (+, *, / only: the same bits on every IEEE machine, which
    libm's and numpy's vectorised cos/sin do not promise)"""
    c, s, term, k = 1.0, theta, theta, 1
    t2 = theta * theta
    cterm = 1.0
    for k in range(1, 16):
        cterm = -cterm * t2 / ((2 * k - 1) * (2 * k))
        c += cterm
        term = -term * t2 / ((2 * k) * (2 * k + 1))
        s += term
    return c, s


def channel_exact(iq, N, cfo=0.0, echoes=(), snr_db=None, seed=5, ref_span=(1000, 101000)):
    """channel() with bit-reproducible arithmetic, for the committed golden fixtures (tests/golden/): every operation is a single IEEE double
    add / subtract / multiply on real arrays (no complex ufunc, no libm, nothing a SIMD dispatch could fuse or reorder); the carrier offset's
    phasor comes from a two-level table of powers built by sequential multiplication.  Same model as channel()."""
    x = np.asarray(iq)
    xr, xi = x.real.astype(np.float64), x.imag.astype(np.float64)
    for d, a in echoes:
        ar, ai = float(np.real(a)), float(np.imag(a))
        yr, yi = xr.copy(), xi.copy()
        t = xr[:-d] * ar; u = xi[:-d] * ai; yr[d:] = yr[d:] + (t - u)
        t = xr[:-d] * ai; u = xi[:-d] * ar; yi[d:] = yi[d:] + (t + u)
        xr, xi = yr, yi
    if cfo:
        theta = 2.0 * 3.141592653589793 * cfo / N
        n = len(xr)
        B = 4096
        wc, ws = _cos_sin_exact(theta)
        ac, as_ = np.empty(B), np.empty(B)
        c, s_ = 1.0, 0.0
        for i in range(B):                                           # w^i, renormalised by nothing: 4096 steps lose ~1e-13
            ac[i], as_[i] = c, s_
            c, s_ = c * wc - s_ * ws, c * ws + s_ * wc
        bc_, bs_ = c, s_                                             # w^4096
        nb = (n + B - 1) // B
        kc, ks = np.empty(nb), np.empty(nb)
        c, s_ = 1.0, 0.0
        for k in range(nb):
            kc[k], ks[k] = c, s_
            c, s_ = c * bc_ - s_ * bs_, c * bs_ + s_ * bc_
        idx = np.arange(n)
        hi, lo = idx >> 12, idx & (B - 1)
        t = kc[hi] * ac[lo]; u = ks[hi] * as_[lo]; pr = t - u
        t = kc[hi] * as_[lo]; u = ks[hi] * ac[lo]; pi_ = t + u
        t = xr * pr; u = xi * pi_; yr = t - u
        t = xr * pi_; u = xi * pr; yi = t + u
        xr, xi = yr, yi
    if snr_db is not None:
        rng = np.random.RandomState(seed)
        a, b = ref_span
        t = xr[a:b] * xr[a:b]; u = xi[a:b] * xi[a:b]
        p = float(np.cumsum(t + u)[-1]) / (b - a)                    # cumsum: one fixed order of additions (a pairwise np.sum may regroup)
        sig = (p / (10.0 ** (snr_db / 10.0)) / 2.0) ** 0.5
        xr = xr + sig * rng.randn(len(xr)); xi = xi + sig * rng.randn(len(xi))
    out = np.empty(len(xr), np.complex64)
    out.real = xr.astype(np.float32); out.imag = xi.astype(np.float32)
    return out


def tx_scale(c):
    """TX multiply_const * RX multiply_const of the demo flowgraphs (apps/dvbt_{tx,rx}_demo*.grc)."""
    return float(np.float32(0.0022097087) * np.float32(0.0022097087 if c.mode == T2k else 0.00055242272))


def tx(c, ts, scale=None, lead_in=0, tail=0, want_freq=False, packet0=0):
    """TS bytes -> complex64 baseband at the 64/7 Msps tap. lead_in/tail: zero samples added.
    packet0: index of ts[0] in the whole TS (o_tx_generate_from)."""
    L = lib()
    npk = len(ts) // 188
    nsym = L.o_tx_symbols_for_packets(C.byref(c), npk)
    n = nsym * (c.N + c.cp)
    iq = np.zeros(lead_in + n + tail, dtype=np.complex64)
    freq = np.zeros((nsym, c.N), dtype=np.complex64) if want_freq else None
    if scale is None:
        scale = tx_scale(c)
    body = iq[lead_in:lead_in + n]
    w = L.o_tx_generate_from(C.byref(c), _p(ts), npk, packet0, C.c_float(scale), _p(body), n,
                             _p(freq) if want_freq else None)
    assert w == n
    return (iq, freq) if want_freq else iq


def rx(c, iq, snr_db=30.0, bsize=768, rs_compat=0, want=("rs",), max_sym_taps=None, sym_off=0):
    """Run the whole oracle chain; returns dict of requested taps + metadata.
    sym_off > 0: the segment continues a cut stream (o_rx_run_cut)."""
    L = lib()
    iq = np.ascontiguousarray(iq, dtype=np.complex64)
    nsym = len(iq) // (c.N + c.cp) + 2
    ns = nsym if max_sym_taps is None else min(nsym, max_sym_taps)
    t = Taps()
    bufs = {}

    def alloc(name, shape, dtype):
        bufs[name] = np.zeros(shape, dtype=dtype)
        return _p(bufs[name])
    if "acq" in want:
        t.acq_out = alloc("acq", (ns, c.N), np.complex64); t.acq_cap = ns
    if "fft" in want:
        t.fft_out = alloc("fft", (ns, c.N), np.complex64); t.fft_cap = ns
    if "eq" in want:
        t.eq_out = alloc("eq", (ns, c.payload), np.complex64); t.eq_cap = ns
    if "demap" in want:
        t.demap_out = alloc("demap", (ns, c.payload), np.uint8)
    if "symdeint" in want:
        t.symdeint_out = alloc("symdeint", (ns, c.payload), np.uint8)
    if "bitdeint" in want:
        t.bitdeint_out = alloc("bitdeint", (ns, c.payload), np.uint8)
    if "bitdeint_lp" in want:
        t.bitdeint_lp_out = alloc("bitdeint_lp", (ns, c.payload), np.uint8)
    t.sym_cap = ns
    vitcap = nsym * c.payload * c.m * c.k // (8 * c.n) + 64
    if "vit" in want:
        t.vit_out = alloc("vit", (vitcap,), np.uint8); t.vit_cap = vitcap
    if "deint" in want:
        t.deint_out = alloc("deint", (vitcap,), np.uint8); t.deint_cap = vitcap
    if "rs" in want:
        t.rs_out = alloc("rs", (vitcap,), np.uint8); t.rs_cap = vitcap
    if "ts" in want:
        t.ts_out = alloc("ts", (vitcap,), np.uint8); t.ts_cap = vitcap
    t.cp_start = alloc("cp_start", (nsym,), np.int32)
    t.epsilon = alloc("epsilon", (nsym,), np.float32)
    t.sym_index = alloc("sym_index", (nsym,), np.int32)
    t.freq_offset = alloc("freq_offset", (nsym,), np.int32)
    t.call_pos = alloc("call_pos", (nsym,), np.int64)
    t.sync_flag = alloc("sync_flag", (nsym,), np.uint8)
    t.sf_flag = alloc("sf_flag", (nsym,), np.uint8)
    t.meta_cap = nsym
    trunc = L.o_rx_run_cut(C.byref(c), _p(iq), len(iq), C.c_float(snr_db), bsize, rs_compat, sym_off, C.byref(t))
    out = {"truncated": trunc, "n_acquired": t.n_acquired, "first_out_symbol": t.first_out_symbol,
           "rs_fail": t.rs_fail, "rs_corr": t.rs_corr, "t_stage": list(t.t_stage),
           "ts_first_packet": t.ts_first_packet, "stream_rs_items": t.stream_rs_items, "stream_symbol_offset": sym_off}
    for k, n in (("acq", t.acq_n), ("fft", t.fft_n), ("eq", t.eq_n), ("demap", t.sym_n),
                 ("symdeint", t.sym_n), ("bitdeint", t.sym_n), ("bitdeint_lp", t.sym_n), ("vit", t.vit_n), ("deint", t.deint_n),
                 ("rs", t.rs_n), ("ts", t.ts_n)):
        if k in bufs:
            out[k] = bufs[k][:n]
    for k in ("cp_start", "epsilon", "sym_index", "freq_offset", "call_pos", "sync_flag", "sf_flag"):
        out[k] = bufs[k][:t.n_acquired]
    starts = np.flatnonzero(out["sync_flag"])
    out["lock_periods"] = [(int(out["call_pos"][a]), int(b - a)) for a, b in zip(starts, list(starts[1:]) + [t.n_acquired])]
    return out


def resampler_taps(interp, decim):
    """(taps, reduced interp, reduced decim) of rational_resampler_ccc(interp, decim, taps=None, fbw=None)"""
    L = lib()
    ri, rd = C.c_int(), C.c_int()
    n = L.o_resampler_design(interp, decim, C.byref(ri), C.byref(rd), None, 0)
    t = np.zeros(n, np.float32)
    L.o_resampler_design(interp, decim, C.byref(ri), C.byref(rd), _p(t), n)
    return t, ri.value, rd.value


def resample(x, interp, decim, scale=1.0):
    """rational_resampler_ccc(interp, decim) + multiply_const(scale) over a whole stream (oracle/o_resample.c)"""
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.complex64)
    no = L.o_resampler_nout(interp, decim, len(x))
    y = np.zeros(no, np.complex64)
    m = L.o_resample_scale(interp, decim, C.c_float(scale), _p(x), len(x), _p(y), no)
    return y[:m]
