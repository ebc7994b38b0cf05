"""Known-answer tests pinning the oracle's table construction.

The values are the ones SURVEY.md Appendix F records from the *compiled reference*
(lib/symbol_inner_interleaver_impl.cc, lib/reed_solomon.cc, lib/dvbt_demap_impl.cc,
lib/reference_signals_impl.cc run against mock GNU Radio headers during the survey).
The reference itself holds no tests or fixtures for these stages (all qa_* are stubs).
"""
import ctypes as C
import numpy as np


def test_symbol_permutation_H(po):
    L = po.lib()
    for mode, payload, first, last, s1, s2 in (
            (po.T2k, 1512, [0, 1024, 16, 1025, 128, 1056, 2, 1280, 4, 1088, 513, 1160], 1032,
             1142316, 865154924),
            (po.T8k, 6048, [0, 4096, 128, 4128, 2048, 4104, 1, 5120, 256, 4192, 2560, 4140], 4226,
             18286128, 55245142968)):
        c = po.cfg(po.QAM16, po.C1_2, mode)
        h = np.zeros(payload, np.int32)
        L.o_sym_H(C.byref(c), h.ctypes.data_as(C.c_void_p))
        assert list(h[:12]) == first
        assert h[-1] == last
        assert int(h.astype(np.int64).sum()) == s1
        assert int(((np.arange(payload, dtype=np.int64) + 1) * h).sum()) == s2
        assert sorted(h) == list(range(payload))


def test_rs_generator_and_ramp_parity(po):
    L = po.lib()
    rs = po.RS()
    L.o_rs_init(C.byref(rs))
    msg = np.zeros(239, np.uint8)
    msg[238] = 1
    par = np.zeros(16, np.uint8)
    L.o_rs_encode(C.byref(rs), msg.ctypes.data_as(C.c_void_p), par.ctypes.data_as(C.c_void_p))
    assert list(par) == [59, 13, 104, 189, 68, 209, 30, 8, 163, 65, 41, 229, 98, 50, 36, 59]
    msg = np.zeros(239, np.uint8)
    msg[51:] = np.arange(188)
    L.o_rs_encode(C.byref(rs), msg.ctypes.data_as(C.c_void_p), par.ctypes.data_as(C.c_void_p))
    assert bytes(par).hex() == "311d78d6c860f878b7189f1a54961d5f"


def test_rs_decode_capability_and_compat_quirk(po):
    """compat=0: every <=8-error pattern corrected, 9+ not (SURVEY App. F last item, 'omega
    enlarged'); compat=1: single error at index>=1 returns 1 but stays wrong (as compiled)."""
    L = po.lib()
    rs = po.RS()
    L.o_rs_init(C.byref(rs))
    rng = np.random.RandomState(7)
    for nerr in (0, 1, 4, 8, 9, 10):
        for _ in range(40):
            w = np.zeros(255, np.uint8)
            w[51:239] = rng.randint(0, 256, 188)
            par = np.zeros(16, np.uint8)
            L.o_rs_encode(C.byref(rs), w.ctypes.data_as(C.c_void_p), par.ctypes.data_as(C.c_void_p))
            w[239:] = par
            good = w.copy()
            pos = rng.choice(np.arange(51, 255), nerr, replace=False)
            for p in pos:
                w[p] ^= rng.randint(1, 256)
            r = L.o_rs_decode(C.byref(rs), w.ctypes.data_as(C.c_void_p), 0)
            if nerr <= 8:
                assert r == nerr and (w == good).all()
            else:
                assert not (w == good).all()
    # the as-compiled quirk
    w = np.zeros(255, np.uint8)
    w[51:239] = rng.randint(0, 256, 188)
    par = np.zeros(16, np.uint8)
    L.o_rs_encode(C.byref(rs), w.ctypes.data_as(C.c_void_p), par.ctypes.data_as(C.c_void_p))
    w[239:] = par
    good = w.copy()
    w[100] ^= 0x5a
    r = L.o_rs_decode(C.byref(rs), w.ctypes.data_as(C.c_void_p), 1)
    assert r == 1 and w[100] != good[100] and w[0] == 0x5a


def test_constellation_labels(po):
    L = po.lib()
    c = po.cfg(po.QAM64, po.C7_8, po.T8k)
    pts = np.zeros(64, np.complex64)
    L.o_constellation(C.byref(c), C.c_float(1.0), pts.ctypes.data_as(C.c_void_p))
    pts = pts / c.norm
    exp = {0x00: (7, 7), 0x01: (7, 5), 0x05: (7, 3), 0x04: (7, 1), 0x02: (5, 7), 0x0a: (3, 7),
           0x08: (1, 7), 0x10: (7, -7), 0x20: (-7, 7), 0x3c: (-1, -1)}
    for k, (re, im) in exp.items():
        assert abs(pts[k] - complex(re, im)) < 1e-5
    c = po.cfg(po.QAM16, po.C1_2, po.T2k)
    pts = np.zeros(16, np.complex64)
    L.o_constellation(C.byref(c), C.c_float(1.0), pts.ctypes.data_as(C.c_void_p))
    pts = pts / c.norm
    for k, (re, im) in {1: (3, 1), 2: (1, 3), 4: (3, -3), 8: (-3, 3)}.items():
        assert abs(pts[k] - complex(re, im)) < 1e-5
    assert abs(c.norm - 1 / np.sqrt(10)) < 1e-7


def test_pilots_8k_first_carriers(po):
    c = po.cfg(po.QAM64, po.C7_8, po.T8k)
    ts = po.make_ts(2000, 3)
    _, freq = po.tx(c, ts, scale=1.0, want_freq=True)
    zl = c.zeros_left
    exp = [
        {0: -4/3, 12: 4/3, 24: 4/3, 34: 1, 36: 4/3, 48: -4/3, 50: -1, 54: -4/3, 60: 4/3, 72: 4/3, 84: -4/3, 87: 4/3, 96: -4/3, 108: -4/3},
        {0: -4/3, 3: -4/3, 15: 4/3, 27: 4/3, 34: 1, 39: -4/3, 48: -4/3, 50: -1, 51: -4/3, 54: -4/3, 63: 4/3, 75: 4/3, 87: 4/3, 99: 4/3},
        {0: -4/3, 6: -4/3, 18: 4/3, 30: -4/3, 34: 1, 42: -4/3, 48: -4/3, 50: -1, 54: -4/3, 66: -4/3, 78: -4/3, 87: 4/3, 90: -4/3, 102: -4/3},
        {0: -4/3, 9: -4/3, 21: -4/3, 33: 4/3, 34: -1, 45: 4/3, 48: -4/3, 50: 1, 54: -4/3, 57: -4/3, 69: 4/3, 81: 4/3, 87: 4/3, 93: 4/3}]
    for s in range(4):
        for k, v in exp[s].items():
            assert abs(freq[s, zl + k] - v) < 1e-6, (s, k)
    assert c.zeros_left == 688 and c.payload == 6048
    # payload carriers (Appendix F): everything that is neither scattered / continual pilot nor TPS carrier, ascending
    # (reference_signals_impl.cc:1073-1106); in the generator's spectrum they are the bins with a non-zero imaginary part
    first = [[1, 2, 3, 4, 5, 6], [1, 2, 4, 5, 6, 7], [1, 2, 3, 4, 5, 7], [1, 2, 3, 4, 5, 6]]
    for s in range(4):
        pay = [k for k in range(c.Kmax + 1) if abs(freq[s, zl + k].imag) > 1e-3]
        assert pay[:6] == first[s] and pay[-1] == 6815 and len(pay) == 6048, (s, pay[:8], pay[-1], len(pay))


def test_tps_bits_frame0(po):
    L = po.lib()
    c = po.cfg(po.QAM64, po.C7_8, po.T8k)
    wk = np.zeros(c.Kmax + 1, np.int8)
    L.o_prbs_wk(C.byref(c), wk.ctypes.data_as(C.c_void_p))
    t = np.zeros(68, np.uint8)
    L.o_tps_format(C.byref(c), 0, wk.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p))
    exp = "0011010111101110" "010111" "00" "10" "000" "100" "100" "00" "01" "00000000" "000000" "01001111110000"
    assert "".join(str(b) for b in t[1:]) == exp
    assert L.o_bch_check(t.ctypes.data_as(C.c_void_p)) == 0
    t[30] ^= 1
    assert L.o_bch_check(t.ctypes.data_as(C.c_void_p)) == -1


def test_config_dimensions(po):
    for const, cr, mode, exp in ((po.QAM16, po.C1_2, po.T2k, (2048, 64, 1705, 172, 1512, 4, 1, 2)),
                                 (po.QAM64, po.C7_8, po.T8k, (8192, 256, 6817, 688, 6048, 6, 7, 8)),
                                 (po.QPSK, po.C7_8, po.T8k, (8192, 256, 6817, 688, 6048, 2, 7, 8))):
        c = po.cfg(const, cr, mode)
        assert (c.N, c.cp, c.Kmax + 1, c.zeros_left, c.payload, c.m, c.k, c.n) == exp
