"""The arithmetic k_soft4.hpp rests on, as a pure-Python model (no GPU): the in-place trellis of the hard kernel's cell layout (cell c holds state rotl6(c, u mod 6)
at relative step u; the butterfly partner differs in cell bit 5 - u mod 6), correlation branch metrics picked by class out of [A, B, -B, -A], one decision bit per
cell and step (1 = the survivor came from the partner), and the traceback in PHYSICAL coordinates z = lane-in-row | h << 6 | r << 7 where the partner is an XOR
with a constant per phase and the decoded bit is one (or the XOR of two) bits of z.  Encoder: inner_coder_impl.cc:33-48 (r = state | b << 6, X = parity(r & 0x79),
Y = parity(r & 0x5b), next state = r >> 1); the cell layout works in the bit-reversed state convention of d_viterbi.c (polynomials 0x4f, 0x6d)."""
import random

import pytest


def par(x):
    return bin(x).count("1") & 1


def rotl6(c, p):
    p %= 6
    return ((c << p) | (c >> (6 - p))) & 63


def phys(a):      # v3_phys: logical a3 a2 a1 a0 -> physical lane-in-row (a2 <-> row_half_mirror)
    a2 = (a >> 2) & 1
    return (a & 8) | (a2 << 2) | ((a & 3) ^ (3 if a2 else 0))


def logical(p):   # v3_log
    q = p & 7
    a2 = (q >> 2) & 1
    return (p & 8) | (a2 << 2) | ((q ^ (7 if a2 else 0)) & 3)


def z_of(c):
    return phys(c & 15) | (((c >> 4) & 1) << 6) | (((c >> 5) & 1) << 7)


def c_of(z):
    return (((z >> 7) & 1) << 5) | (((z >> 6) & 1) << 4) | logical(z & 15)


FLIP = (0x80, 0x40, 8, 7, 2, 1)     # phase 0..5: VGPR swap, half swap, row_ror:8, row_half_mirror, quad_perm[2,3,0,1], quad_perm[1,0,3,2]


def decoded_bit(z, p1):
    """LSB of the state after the step = cell bit (6 - p1) % 6 = a0, r, h, a3, a2, a1 for p1 = 0..5, read off the physical cell"""
    return [(z ^ (z >> 2)) & 1, (z >> 7) & 1, (z >> 6) & 1, (z >> 3) & 1, (z >> 2) & 1, ((z >> 1) ^ (z >> 2)) & 1][p1]


@pytest.mark.parametrize("seed,noise", [(1, 0), (2, 5), (3, 7)])
def test_in_place_soft_decoder_and_physical_traceback(seed, noise):
    rnd = random.Random(seed)
    n = 700
    msg = [rnd.randint(0, 1) for _ in range(n)]
    s, sx, sy = 0, [], []
    for b in msg:
        r = s | (b << 6)
        x, y = par(r & 0x79), par(r & 0x5b)
        s = r >> 1
        sx.append((8 if x == 0 else -8) + rnd.randint(-noise, noise))
        sy.append((8 if y == 0 else -8) + rnd.randint(-noise, noise))
    v = [0] * 64
    dec = []
    for u in range(n):
        p = u % 6
        a, b = sx[u] + sy[u], sy[u] - sx[u]
        xs, ys = [0] * 64, [0] * 64
        for c in range(64):
            i = rotl6(c, p) & 31
            c0 = ((i >> 2) ^ (i >> 1) ^ i) & 1                  # parity(2i & 0x4f)
            c1 = ((i >> 4) ^ (i >> 2) ^ (i >> 1)) & 1           # parity(2i & 0x6d)
            d = (a, b, -b, -a)[c0 | (c1 << 1)]
            xs[c], ys[c] = v[c] + d, v[c] - d
        d_u = [1 if ys[c ^ (1 << (5 - p))] > xs[c] else 0 for c in range(64)]
        v = [max(xs[c], ys[c ^ (1 << (5 - p))]) for c in range(64)]
        dec.append(d_u)
    z = z_of(max(range(64), key=lambda c: v[c]))
    out = [0] * n
    for u in range(n - 1, -1, -1):
        p, p1 = u % 6, (u + 1) % 6
        c = c_of(z)
        out[u] = decoded_bit(z, p1)
        assert out[u] == rotl6(c, p1) & 1
        if dec[u][c]:
            z ^= FLIP[p]
            assert c_of(z) == c ^ (1 << (5 - p))                # the XOR in physical coordinates is the logical butterfly partner
    assert out[:n - 48] == msg[:n - 48]


def test_physical_lane_map_is_a_bijection():
    assert sorted(phys(a) for a in range(16)) == list(range(16)) and all(logical(phys(a)) == a for a in range(16))
