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


def test_energy_dispersal_prbs_is_the_standards(po):
    """ETSI EN 300 744 4.3.1: PRBS 1 + X^14 + X^15, register loaded with 100101010000000 at the first (inverted, 0xB8) sync byte of every group of 8
    packets, its first output bit applied to the first bit behind that sync byte; it keeps running under the other seven sync bytes without being applied.
    An LFSR written from that text alone (not from the reference, not from the oracle) must reproduce what the oracle's chain carries behind its RS decoder
    for a TS of zero payload -- and its first bytes are the well-known 0x03 0xF6 0x08 0x34 0x30 0xB8 0xA3 0x93."""
    def prbs_bytes(n):
        reg = [1, 0, 0, 1, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0]         # stages 1..15
        out = []
        for _ in range(n):
            b = 0
            for _ in range(8):
                bit = reg[13] ^ reg[14]
                reg = [bit] + reg[:14]
                b = (b << 1) | bit
            out.append(b)
        return out
    seq = prbs_bytes(8 * 188 - 1)                                      # one group of 8 packets behind its first sync byte
    assert seq[:8] == [0x03, 0xF6, 0x08, 0x34, 0x30, 0xB8, 0xA3, 0x93]
    group = [0xB8] + seq[:187]
    for k in range(1, 8):
        group += [0x47] + seq[188 * k:188 * k + 187]                   # the PRBS runs on under the sync byte (one byte of it is skipped), not applied
    group = np.array(group, np.uint8)
    c = po.cfg(po.QAM16, po.C1_2, po.T2k)
    npk = po.packets_per_superframe(c) * 3
    ts = np.zeros(npk * 188, np.uint8)
    ts[0::188] = 0x47
    o = po.rx(c, po.tx(c, ts, lead_in=500, tail=3 * c.N), want=("rs", "ts"))
    rs = o["rs"].reshape(-1, 188)
    starts = [i for i in range(len(rs) - 8) if rs[i, 0] == 0xB8 and i >= 11]
    assert len(starts) >= 10 and all(b - a == 8 for a, b in zip(starts, starts[1:]))
    for i in starts[:10]:
        assert (rs[i:i + 8].reshape(-1) == group).all()
    assert (o["ts"].reshape(-1, 188)[:, 1:] == 0).all() and (o["ts"][0::188] == 0x47).all()


def test_inner_code_is_the_standards(po):
    """ETSI EN 300 744 4.3.3: mother code G1 = 171 oct (X), G2 = 133 oct (Y), puncturing patterns of table 4 with the transmitted sequence X1 Y1 | X1 Y1 Y2 |
    X1 Y1 Y2 X3 | X1 Y1 Y2 X3 Y4 X5 | X1 Y1 Y2 Y3 Y4 X5 Y6 X7.  An encoder written from that text alone, fed with the bytes the oracle's Viterbi decoder
    delivers on a clean loopback, must reproduce the coded bits the oracle's chain carries INTO that decoder (bit de-interleaver tap, m bits per item, MSB
    first): polynomials, puncturing phase at the superframe start and bit order in one known answer per code rate."""
    seqs = {po.C1_2: ("X1", "Y1"), po.C2_3: ("X1", "Y1", "Y2"), po.C3_4: ("X1", "Y1", "Y2", "X3"), po.C5_6: ("X1", "Y1", "Y2", "X3", "Y4", "X5"),
            po.C7_8: ("X1", "Y1", "Y2", "Y3", "Y4", "X5", "Y6", "X7")}
    for cr, seq in seqs.items():
        c = po.cfg(po.QAM16, cr, po.T2k)
        ts = po.make_ts(po.packets_per_superframe(c) * 3, 5)
        o = po.rx(c, po.tx(c, ts, lead_in=500, tail=3 * c.N), want=("bitdeint", "vit"))
        info = np.unpackbits(o["vit"])
        k = max(int(s[1]) for s in seq)                               # input bits per puncturing period
        d = np.concatenate([np.zeros(6, np.uint8), info])            # the register in front of the first delivered bit: unknown, the first steps are skipped
        x = d[6:] ^ d[5:-1] ^ d[4:-2] ^ d[3:-3] ^ d[:-6]              # 171 oct = 1111001: taps at delays 0, 1, 2, 3, 6
        y = d[6:] ^ d[4:-2] ^ d[3:-3] ^ d[1:-5] ^ d[:-6]              # 133 oct = 1011011: taps at delays 0, 2, 3, 5, 6
        n = (len(info) // k) * k
        coded = np.stack([(x if s[0] == "X" else y)[int(s[1]) - 1:n:k] for s in seq], axis=1).reshape(-1)
        rx_bits = np.unpackbits(o["bitdeint"].reshape(-1, 1), axis=1)[:, 8 - c.m:].reshape(-1)
        lo, hi = 8 * len(seq), min(len(coded), len(rx_bits)) - 64
        assert hi > 100000 and (coded[lo:hi] == rx_bits[lo:hi]).all(), cr


def test_outer_deinterleaver_is_the_standards(po):
    """ETSI EN 300 744 4.3.2 (Forney, I = 12, M = 17): the byte at position n of the interleaved stream travels on branch n mod 12; the de-interleaver's branch b
    delays it by 17 (11 - b) cells of 12 bytes.  Written from that text: deint[n] = vit[n - 204 (11 - n mod 12)] (zero fill in front), counted from the first
    byte behind a superframe start (a packet start, branch 0).  And the RS code of the words that come out: 16 parity bytes make the remainder of the
    shortened (204, 188) code vanish -- every word behind the fill is a codeword of g(x) = prod (x + alpha^i), i = 0 .. 15, over GF(256) with p(x) = 0x11d."""
    c = po.cfg(po.QAM64, po.C2_3, po.T2k)
    ts = po.make_ts(po.packets_per_superframe(c) * 3, 11)
    o = po.rx(c, po.tx(c, ts, lead_in=300, tail=3 * c.N), want=("vit", "deint"))
    vit, de = o["vit"], o["deint"]
    n = np.arange(len(de))
    src = n - 204 * (11 - n % 12)
    want = np.where(src >= 0, vit[np.clip(src, 0, len(vit) - 1)], 0).astype(np.uint8)
    ok = src < len(vit)
    assert len(de) > 50000 and (de[ok] == want[ok]).all()
    # GF(256) from the text: alpha = 2, p(x) = x^8 + x^4 + x^3 + x^2 + 1
    exp = np.zeros(512, np.int64); log = np.zeros(256, np.int64)
    v = 1
    for i in range(255):
        exp[i] = v; log[v] = i
        v <<= 1
        if v & 0x100: v ^= 0x11d
    exp[255:510] = exp[:255]
    words = de[204 * 11:(len(de) // 204) * 204].reshape(-1, 204).astype(np.int64)
    for root in range(16):                                            # a codeword vanishes at alpha^0 .. alpha^15
        acc = np.zeros(len(words), np.int64)
        for j in range(204):                                          # Horner, first byte = highest power
            nz = acc != 0
            acc = np.where(nz, exp[(log[acc] + root) % 255], 0) ^ words[:, j]
        assert (acc == 0).all(), root


def test_inner_interleavers_are_the_standards(po):
    """ETSI EN 300 744 4.3.4.  Symbol interleaver: H(q) from the shift register R'_i (2k: new bit R'[0] ^ R'[3]; 8k: R'[0] ^ R'[1] ^ R'[4] ^ R'[6]), the bit
    permutations of tables 3a / 3b and the alternating top bit, written from that text.  Bit interleaver: demultiplexing x_di -> b_[di mod v div (v/2) +
    2 (di mod v/2)], di div v and the six interleavers H_e(w) = (w + {0, 63, 105, 42, 21, 84}) mod 126 applied to what the oracle's chain feeds its Viterbi
    decoder must give the words that leave its symbol de-interleaver."""
    def etsi_H(mode):
        nr, nmax, perm = (11, 1512, [0, 7, 5, 1, 8, 2, 6, 9, 3, 4]) if mode == po.T2k else (13, 6048, [5, 11, 3, 0, 10, 8, 6, 9, 2, 4, 1, 7])
        nb, out, rp = nr - 1, [], 0
        for i in range(1 << nr):
            if i < 2:
                rp = 0
            elif i == 2:
                rp = 1
            else:
                new = (rp ^ (rp >> 3)) & 1 if mode == po.T2k else (rp ^ (rp >> 1) ^ (rp >> 4) ^ (rp >> 6)) & 1
                rp = (rp >> 1) | (new << (nb - 1))
            r = 0
            for k in range(nb):                                       # table 3: R' bit nb-1-k goes to R bit perm[k]
                r |= ((rp >> (nb - 1 - k)) & 1) << perm[k]
            h = ((i & 1) << (nr - 1)) + r
            if h < nmax:
                out.append(h)
        return np.array(out)
    L = po.lib()
    for mode, payload in ((po.T2k, 1512), (po.T8k, 6048)):
        c = po.cfg(po.QAM16, po.C1_2, mode)
        h = np.zeros(payload, np.int32)
        L.o_sym_H(C.byref(c), h.ctypes.data_as(C.c_void_p))
        assert (etsi_H(mode) == h).all()
    off = [0, 63, 105, 42, 21, 84]
    for const in (po.QPSK, po.QAM16, po.QAM64):
        c = po.cfg(const, po.C3_4, po.T2k)
        v = c.m
        ts = po.make_ts(po.packets_per_superframe(c) * 2, 3)
        o = po.rx(c, po.tx(c, ts, lead_in=200, tail=3 * c.N), want=("symdeint", "bitdeint"))
        x = np.unpackbits(o["bitdeint"].reshape(-1, 1), axis=1)[:, 8 - v:].reshape(-1)      # the coded bits x_di in order
        nblk = min(len(x) // (126 * v), 2000)
        x = x[:nblk * 126 * v].reshape(nblk, 126, v)                   # [block][do][di mod v]
        y = np.zeros((nblk, 126), np.int64)
        for k in range(v):
            e = (k % v) // (v // 2) + 2 * (k % (v // 2))
            w = np.arange(126)
            y |= x[:, (w + off[e]) % 126, k].astype(np.int64) << (v - 1 - e)           # a_{e,w} = b_{e,H_e(w)}, a_0 = MSB of the word
        assert (y.reshape(-1) == o["symdeint"].reshape(-1)[:nblk * 126]).all(), const


def test_pilot_sequence_and_positions_are_the_standards(po):
    """ETSI EN 300 744 4.5.2 / 4.5.3 / 4.6: reference sequence w_k from the PRBS X^11 + X^2 + 1 (eleven ones loaded, one bit per carrier from k = 0: taps
    behind delays 9 and 11), pilots at 4/3 (1 - 2 w_k); scattered pilots at k = 3 (l mod 4) + 12 p; the continual pilots and TPS carriers of tables 7 and
    8 (2k lists).  Written from that text; checked against the spectrum the oracle's generator builds and its w_k table."""
    s, w = [1] * 11, []
    for _ in range(6817):
        w.append(s[10])
        s = [s[8] ^ s[10]] + s[:10]
    w = np.array(w)
    L = po.lib()
    c8 = po.cfg(po.QAM64, po.C7_8, po.T8k)
    wk = np.zeros(c8.Kmax + 1, np.int8)
    L.o_prbs_wk(C.byref(c8), wk.ctypes.data_as(C.c_void_p))
    assert (wk == w).all()
    cont = [0, 48, 54, 87, 141, 156, 192, 201, 255, 279, 282, 333, 432, 450, 483, 525, 531, 618, 636, 714, 759, 765, 780, 804, 873, 888, 918, 939, 942, 969,
            984, 1050, 1101, 1107, 1110, 1137, 1140, 1146, 1206, 1269, 1323, 1377, 1491, 1683, 1704]
    tps = [34, 50, 209, 346, 413, 569, 595, 688, 790, 901, 1073, 1219, 1262, 1286, 1469, 1594, 1687]
    c = po.cfg(po.QAM16, po.C1_2, po.T2k)
    _, freq = po.tx(c, po.make_ts(600, 3), scale=1.0, want_freq=True)
    zl = c.zeros_left
    for l in range(8):
        row = freq[l, zl:zl + c.Kmax + 1]
        pil = set(cont) | set(range(3 * (l % 4), c.Kmax + 1, 12))
        for k in sorted(pil):
            assert abs(row[k] - (4.0 / 3.0) * (1 - 2 * w[k])) < 1e-6, (l, k)
        # TPS carriers: DBPSK, real, magnitude 1, the same bit on all of them up to the sign of w_k
        t = np.array([row[k].real * (1 - 2 * w[k]) for k in tps])
        assert np.allclose(np.abs(t), 1.0, atol=1e-6) and np.allclose([row[k].imag for k in tps], 0.0, atol=1e-6) and abs(t.sum()) == len(tps)
        # everything else carries data: not on the real axis at pilot power
        data = [k for k in range(c.Kmax + 1) if k not in pil and k not in tps]
        assert len(data) == 1512 and min(abs(row[k].imag) for k in data) > 1e-3


def test_constellations_are_gray_mapped_as_in_the_standard(po):
    """ETSI EN 300 744 4.3.5, figures 9: y0 is the sign of I and y1 the sign of Q (0 = positive); neighbouring points differ in exactly one bit; the levels are
    the odd integers (uniform) or alpha + 2 i (hierarchical, alpha = 2, 4), normalised by the table's factor."""
    L = po.lib()
    for const, hier in ((po.QPSK, 0), (po.QAM16, 0), (po.QAM64, 0), (po.QAM16, 2), (po.QAM64, 3)):
        c = po.cfg(const, po.C1_2, po.T2k, hierarchy=hier)
        m = c.m
        pts = np.zeros(1 << m, np.complex64)
        L.o_constellation(C.byref(c), C.c_float(1.0), pts.ctypes.data_as(C.c_void_p))
        z = (pts / c.norm).astype(np.complex128)
        lab = np.arange(1 << m)
        assert (((lab >> (m - 1)) & 1) == (z.real < 0)).all() and (((lab >> (m - 2)) & 1) == (z.imag < 0)).all()
        alpha = {0: 1, 1: 1, 2: 2, 3: 4}[hier]
        levels = sorted({round(abs(v), 4) for v in z.real})
        assert levels == [alpha + 2 * i for i in range(1 << (m // 2 - 1))]
        for a in range(1 << m):
            for b in range(a + 1, 1 << m):
                d = z[a] - z[b]
                near = (d.imag == 0 and abs(d.real) == min(2, 2 * alpha) or d.real == 0 and abs(d.imag) == min(2, 2 * alpha)) if alpha == 1 else False
                if near:
                    assert bin(a ^ b).count("1") == 1, (a, b)


def test_tps_frame_is_a_bch_codeword_of_the_standards_generator(po):
    """ETSI EN 300 744 4.6.3: the 53 information bits s1 .. s53 and 14 parity bits of a TPS frame form a word of the BCH (67, 53, t = 2) code shortened from
    (127, 113), generator x^14 + x^9 + x^8 + x^6 + x^5 + x^4 + x^2 + x + 1; 4.6.2: synchronisation word 0011010111101110 in frames 1 and 3, its inverse in
    2 and 4, length indicator 010111 (no cell identification).  Polynomial division written here, not the oracle's checker."""
    L = po.lib()
    g = 0b100001101110111
    for const, cr, mode in ((po.QAM64, po.C7_8, po.T8k), (po.QAM16, po.C1_2, po.T2k), (po.QPSK, po.C3_4, po.T8k)):
        c = po.cfg(const, cr, mode)
        wk = np.zeros(c.Kmax + 1, np.int8)
        L.o_prbs_wk(C.byref(c), wk.ctypes.data_as(C.c_void_p))
        for frame in range(4):
            t = np.zeros(68, np.uint8)
            L.o_tps_format(C.byref(c), frame, wk.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p))
            sync = "".join(str(b) for b in t[1:17])
            assert sync == ("0011010111101110" if frame % 2 == 0 else "1100101000010001") and "".join(str(b) for b in t[17:23]) == "010111"
            assert int(t[23]) * 2 + int(t[24]) == frame                  # frame number, s23 s24
            r = 0
            for b in t[1:68]:                                            # s1 first = highest power
                r = (r << 1) | int(b)
                if r & (1 << 14):
                    r ^= g
            assert r == 0, (const, cr, mode, frame)
