"""The oracle's hierarchical bit de-interleaver (oracle/o_inner.c::o_bit_deinterleave_hier) against a line-by-line transcription of the reference block's loops
(lib/bit_inner_deinterleaver_impl.cc:91-99 d_perm, :138-146 the bit matrix, :160-178 the two outputs), written here with the matrix as the flat array C lays it
out, so that the reference's second indices beyond 125 land where they land in memory.  Bits the reference reads from BEHIND the matrix (64-QAM, row 5) are
undefined there; both sides take 0 and the test counts how many bytes that concerns.  Also: the alpha = 2 / 4 constellations (lib/dvbt_demap_impl.cc:117-165 with
d_alpha of lib/dvbt_config.cc:213-225) against the closed form of ETSI EN 300 744 4.3.5 (levels alpha + 2 j, the normalisation of table 6)."""
import ctypes as C
import numpy as np
import pytest


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _ref_hier(v, x):
    H = lambda e, w: (w + (0, 63, 105, 42, 21, 84)[e]) % 126                       # :34-58
    d_perm = [(i % (v - 2)) // ((v - 2) // 2) + 2 * (i % ((v - 2) // 2)) + 2 for i in range(126 * v)]   # :91-99, hierarchical branch
    outh, outl = np.zeros_like(x), np.zeros_like(x)
    undefined = 0
    for b in range(len(x) // 126):
        flat = np.zeros(v * 126, np.uint8)
        for w in range(126):
            c = int(x[b * 126 + w])
            for e in range(v):
                flat[e * 126 + H(e, w)] = (c >> (v - e - 1)) & 1                   # d_b[e][H(e, w)]
        for i in range(126):
            c = 0
            for k in range(2):
                c = (c << 1) | int(flat[((v * i + k) % 2) * 126 + (v * i + k) // 2])   # d_b[(d_v*i+k) % 2][(d_v*i+k) / 2]
            outh[b * 126 + i] = c
            c = 0
            for k in range(2, v - 2):
                f = d_perm[v * i + k] * 126 + (v * i + k) // (v - 2)               # d_b[d_perm[..]][(d_v*i+k) / (d_v-2)]
                if f >= v * 126:
                    undefined += 1; bit = 0
                else:
                    bit = int(flat[f])
                c = (c << 1) | bit
            outl[b * 126 + i] = c
    return outh, outl, undefined


@pytest.mark.parametrize("const", [1, 2])
def test_hierarchical_bit_deinterleaver_equals_the_reference_loops(po, const):
    c = po.cfg(const, po.C1_2, po.T2k, hierarchy=2)
    rng = np.random.RandomState(5)
    x = rng.randint(0, c.csize, 4 * 126).astype(np.uint8)
    oh, ol = np.zeros_like(x), np.zeros_like(x)
    po.lib().o_bit_deinterleave_hier(C.byref(c), _p(x), _p(oh), _p(ol), C.c_size_t(len(x)))
    rh, rl, undefined = _ref_hier(c.m, x)
    assert (oh == rh).all() and (ol == rl).all()
    assert oh.max() <= 3
    if const == 1:
        assert (ol == 0).all() and undefined == 0        # the LP loop `k = 2; k < d_v - 2` does not run for 16-QAM
    else:
        assert ol.max() == 3 and 0 < undefined < 4 * 126  # 64-QAM: two LP bits; some of row 5's lie behind the matrix


@pytest.mark.parametrize("const,hier,alpha,norm2", [(1, 2, 2, 20), (1, 3, 4, 52), (2, 2, 2, 60), (2, 3, 4, 108), (1, 1, 1, 10), (2, 1, 1, 42)])
def test_hierarchical_constellations(po, const, hier, alpha, norm2):
    c = po.cfg(const, po.C1_2, po.T2k, hierarchy=hier)
    assert c.alpha == alpha and abs(c.norm - 1 / np.sqrt(norm2)) < 1e-7
    pts = np.zeros(c.csize, np.complex64)
    po.lib().o_constellation(C.byref(c), C.c_float(1.0), _p(pts))
    lev = np.unique(np.round(np.abs(pts.real) / c.norm).astype(int))
    assert list(lev) == [alpha + 2 * j for j in range(int(np.sqrt(c.csize)) // 2)]
    assert abs(np.mean(np.abs(pts) ** 2) - 1.0) < 1e-6                              # table 6's factors normalise the mean power
    # the two most significant bits are the quadrant (what the high-priority stream carries)
    for lab, p in enumerate(pts):
        assert (lab >> (c.m - 1)) == (1 if p.real < 0 else 0) and ((lab >> (c.m - 2)) & 1) == (1 if p.imag < 0 else 0)
