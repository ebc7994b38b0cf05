"""The LDS layouts and the FFT index maps of the per-symbol kernels (gr_dvbt_amd/csrc/k_symbol8k.hpp, k_symbol2k.hpp), restated in numpy:
* every swizzle is a bijection of the symbol image;
* every access pattern the kernels' comments claim to be conflict free lands on distinct banks (ds_read_b64: lane groups of 32, bank pair =
  float2 index mod 32; ds_write_b64: lane groups of 16, float2 index mod 16 -- MI355X_MICROARCH.md, LDS);
* the pass decomposition (16 x 16 x (2 x 16) for 8192, 16 x 8 x 16 for 2048, first pass on stride-N/16 samples, last pass written at
  bin ^ N/2) is the shifted forward DFT.
No GPU: these are the design's invariants, the GPU tests check the kernels themselves."""
import numpy as np


def swz1_8k(a):
    return a ^ ((a >> 5) & 15) ^ (((a >> 9) & 1) << 4)


def swz2(b):
    return b ^ ((b >> 4) & 15)


def swz1_2k(k1, idx):
    return k1 * 128 + (idx ^ (k1 & 15) ^ ((k1 & 1) << 4))


def distinct(addr, group, modulus):
    """addr: float2 indices of consecutive lanes; every `group` consecutive lanes must hit `group` different banks (identical addresses broadcast)"""
    addr = np.asarray(addr)
    for g0 in range(0, len(addr), group):
        part = addr[g0:g0 + group]
        uniq = np.unique(part)
        if len(np.unique(uniq % modulus)) != len(uniq):
            return False
    return True


def test_swizzles_are_bijections():
    a = np.arange(8192)
    assert len(np.unique(swz1_8k(a))) == 8192 and swz1_8k(a).max() == 8191
    assert len(np.unique(swz2(a))) == 8192 and swz2(a).max() == 8191
    b = np.arange(2048)
    assert len(np.unique(swz2(b))) == 2048 and swz2(b).max() == 2047
    img = np.array([swz1_2k(k1, idx) for k1 in range(16) for idx in range(128)])
    assert len(np.unique(img)) == 2048 and img.max() == 2047


def test_8k_access_patterns_are_conflict_free():
    tid = np.arange(512)
    for k1 in range(16):                                           # pass 1 store: x[k1 * 512 + (b0 ^ ((k1 & 1) << 4))]
        b0 = tid ^ ((tid >> 5) & 15)
        addr = k1 * 512 + (b0 ^ ((k1 & 1) << 4))
        assert (addr == swz1_8k(k1 * 512 + tid)).all()
        assert distinct(addr, 16, 16)
    k1 = tid >> 5
    mm = (tid & 31) ^ ((k1 & 1) << 4)
    for m1 in range(16):                                           # pass 2 load / store in place
        addr = k1 * 512 + 32 * m1 + (mm ^ m1)
        assert (addr == swz1_8k(k1 * 512 + 32 * m1 + (tid & 31))).all()
        assert distinct(addr, 32, 32) and distinct(addr, 16, 16)
    j1, k1, c = tid & 15, (tid >> 4) & 15, tid >> 8                # pass 3
    jj = j1 ^ ((k1 & 1) << 4)
    rb = k1 * 512 + 32 * j1
    for m2 in range(32):
        addr = rb + (m2 ^ jj)
        assert (addr == swz1_8k(k1 * 512 + 32 * j1 + m2)).all()
        assert distinct(addr, 32, 32)
    base2 = (k1 ^ j1) + 16 * j1 + 256 * c
    for d in range(16):                                            # natural-order store of bin k1 + 16 j1 + 256 c + 512 d at bin ^ 4096
        addr = base2 + 512 * (d ^ 8)
        assert (addr == swz2((k1 + 16 * j1 + 256 * c + 512 * d) ^ 4096)).all()
        assert distinct(addr, 16, 16)
    for b0 in (688, 689, 1000, 4095):                              # the pilot engine reads consecutive carriers: aligned runs are conflict free,
        addr = swz2(b0 + np.arange(64))                            # runs that straddle three 16-carrier blocks cost at most one extra cycle
        for g0 in (0, 32):
            assert np.bincount(addr[g0:g0 + 32] % 32).max() <= (1 if b0 % 16 == 0 else 2)


def test_2k_access_patterns_are_conflict_free():
    t = np.arange(128)
    for k1 in range(16):                                           # pass 1 store
        assert distinct(swz1_2k(k1, t), 16, 16)
    m2 = t & 15
    for h in range(2):                                             # pass 2 (radix 8, two rows per thread)
        k1 = (t >> 4) + 8 * h
        for m1 in range(8):
            addr = swz1_2k(k1, 16 * m1 + m2)
            assert distinct(addr, 32, 32) and distinct(addr, 16, 16)
    k1, j1 = t & 15, t >> 4                                        # pass 3 (radix 16 per row)
    for mm in range(16):
        assert distinct(swz1_2k(k1, 16 * j1 + mm), 32, 32)
    for j2 in range(16):
        assert distinct(swz2(k1 + 16 * j1 + 128 * (j2 ^ 8)), 16, 16)


def _w(n, e):
    return np.exp(-2j * np.pi * np.asarray(e) / n)


def test_8k_pass_structure_is_the_shifted_dft():
    rng = np.random.default_rng(3)
    x = rng.standard_normal(8192) + 1j * rng.standard_normal(8192)
    n2 = np.arange(512)
    y = np.stack([sum(x[n2 + 512 * n1] * _w(16, n1 * k1) for n1 in range(16)) * _w(8192, n2 * k1) for k1 in range(16)])
    m2 = np.arange(32)
    z = np.stack([np.stack([sum(y[k1, 32 * m1 + m2] * _w(16, m1 * j1) for m1 in range(16)) * _w(512, m2 * j1) for j1 in range(16)]) for k1 in range(16)])
    out = np.zeros(8192, complex)
    b = np.arange(16)
    for k1 in range(16):
        for j1 in range(16):
            for c in range(2):
                u = (z[k1, j1, b] + (-1) ** c * z[k1, j1, b + 16]) * _w(32, b * c)
                for d in range(16):
                    out[(k1 + 16 * j1 + 256 * c + 512 * d) ^ 4096] = np.sum(u * _w(16, b * d))
    ref = np.fft.fftshift(np.fft.fft(x))
    assert np.abs(out - ref).max() <= 1e-9 * np.abs(ref).max()


def test_2k_pass_structure_is_the_shifted_dft():
    rng = np.random.default_rng(4)
    x = rng.standard_normal(2048) + 1j * rng.standard_normal(2048)
    n2 = np.arange(128)
    y = np.stack([sum(x[n2 + 128 * n1] * _w(16, n1 * k1) for n1 in range(16)) * _w(2048, n2 * k1) for k1 in range(16)])
    m2 = np.arange(16)
    z = np.stack([np.stack([sum(y[k1, 16 * m1 + m2] * _w(8, m1 * j1) for m1 in range(8)) * _w(128, m2 * j1) for j1 in range(8)]) for k1 in range(16)])
    out = np.zeros(2048, complex)
    for k1 in range(16):
        for j1 in range(8):
            for j2 in range(16):
                out[(k1 + 16 * j1 + 128 * j2) ^ 1024] = np.sum(z[k1, j1] * _w(16, m2 * j2))
    ref = np.fft.fftshift(np.fft.fft(x))
    assert np.abs(out - ref).max() <= 1e-9 * np.abs(ref).max()


def test_demapper_margin_covers_the_float_rounding():
    """s8_demap_cell: t cells inside a boundary the neighbouring level's squared distance is larger by 2 t step^2; the float32 evaluation of
    fl(fl(dx^2) + fl(dy^2)) must order the two points like the exact values whenever t >= the kernel's margin, anywhere within its reach."""
    rng = np.random.default_rng(5)
    margin, reach, n = 2.0e-5, 4.0, 8
    step = np.float32(2.0 / np.sqrt(42.0))
    lev = (np.arange(n) * 2 - (n - 1)).astype(np.float32) * np.float32(1.0 / np.sqrt(42.0))
    for _ in range(20000):
        j = rng.integers(0, n - 1)                                  # boundary between levels j and j + 1
        t = margin * (1 + 3 * rng.random())
        side = rng.integers(0, 2)
        mid = (np.float64(lev[j]) + np.float64(lev[j + 1])) / 2
        x = np.float32(mid + (t if side else -t) * np.float64(step))
        y = np.float32((rng.random() * (n + 2 * reach) - n / 2 - reach) * np.float64(step))
        ky = int(np.argmin(np.abs(lev.astype(np.float64) - np.float64(y))))
        d = []
        for lx in (lev[j], lev[j + 1]):
            dx, dy = np.float32(x - lx), np.float32(y - lev[ky])
            d.append(np.float32(np.float32(dx * dx) + np.float32(dy * dy)))
        assert (d[1] < d[0]) == bool(side) and d[0] != d[1]
