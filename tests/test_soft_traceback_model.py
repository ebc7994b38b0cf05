"""Model of k_soft4.hpp's traceback design (no GPU): the decisions of a soft-input K = 7 decoder are traced back (a) as ONE chain from the best end state and
(b) in segments, each from the best state recorded 192 steps (24 groups of 8) behind its end, as viterbi_soft4_kernel does with 8 segments per decoder.
Near the waterfall, on the unpunctured code and on the 7/8 puncturing pattern (erasures = 0), the two must decode the same bits up to a handful of positions,
and neither may be the better decoder: that is what licenses the 8-fold shorter dependent chains.
Encoder: inner_coder_impl.cc:33-48; puncturing: ETSI EN 300 744 table 4 / viterbi_decoder_impl.cc:95-124 (rate 7/8: X 1000101, Y 1111010)."""
import numpy as np
import pytest


def par(x):
    return bin(x).count("1") & 1


NEXT = np.array([[(s | (b << 6)) >> 1 for b in (0, 1)] for s in range(64)])
OUTX = np.array([[par((s | (b << 6)) & 0x79) for b in (0, 1)] for s in range(64)])
OUTY = np.array([[par((s | (b << 6)) & 0x5b) for b in (0, 1)] for s in range(64)])
# predecessors of state ns: p = ((ns & 31) << 1) | x, input bit = ns >> 5
PRED = np.array([[((ns & 31) << 1) | x for x in (0, 1)] for ns in range(64)])
BIN = np.arange(64) >> 5
SX = np.array([[1 - 2 * par((PRED[ns, x] | (BIN[ns] << 6)) & 0x79) for x in (0, 1)] for ns in range(64)])
SY = np.array([[1 - 2 * par((PRED[ns, x] | (BIN[ns] << 6)) & 0x5b) for x in (0, 1)] for ns in range(64)])


def forward(sx, sy, block=48):
    """add-compare-select over all steps; returns the decisions [step][state] and the best state after every `block` steps"""
    n = len(sx)
    m = np.zeros(64, np.int64)
    dec = np.zeros((n, 64), np.uint8)
    best = {}
    for t in range(n):
        c0 = m[PRED[:, 0]] + sx[t] * SX[:, 0] + sy[t] * SY[:, 0]
        c1 = m[PRED[:, 1]] + sx[t] * SX[:, 1] + sy[t] * SY[:, 1]
        dec[t] = c1 > c0
        m = np.where(c1 > c0, c1, c0)
        if (t + 1) % block == 0:
            m -= m.max()
            best[t + 1] = int(np.argmax(m))
    return dec, best, int(np.argmax(m))


def trace(dec, s, t_hi, t_lo, out):
    """walk from state s after step t_hi - 1 down to step t_lo, writing the input bits of steps [t_lo, t_hi) that the caller keeps"""
    for t in range(t_hi - 1, t_lo - 1, -1):
        out[t] = s >> 5
        s = PRED[s, dec[t, s]]
    return s


@pytest.mark.parametrize("punctured,sigma", [(False, 6.8), (True, 3.5)])
def test_segmented_traceback_equals_the_single_chain(punctured, sigma):
    rng = np.random.RandomState(7)
    n = 48 * 400
    msg = rng.randint(0, 2, n)
    s, sx, sy = 0, np.zeros(n, np.int64), np.zeros(n, np.int64)
    px, py = ([1, 0, 0, 0, 1, 0, 1], [1, 1, 1, 1, 0, 1, 0]) if punctured else ([1], [1])
    for t, b in enumerate(msg):
        x, y = OUTX[s, b], OUTY[s, b]
        s = NEXT[s, b]
        sx[t] = np.clip(np.rint(8 * (1 - 2 * x) + sigma * rng.randn()), -31, 31) * px[t % len(px)]
        sy[t] = np.clip(np.rint(8 * (1 - 2 * y) + sigma * rng.randn()), -31, 31) * py[t % len(py)]
    dec, best, end = forward(sx, sy)
    one = np.zeros(n, np.int64)
    trace(dec, end, n, 0, one)
    seg = np.zeros(n, np.int64)
    S, PRE = 288 * 4, 192                                            # segment length and pre-roll in steps (multiples of the 48-step block)
    for lo in range(0, n, S):
        hi = min(lo + S, n)
        top = min(hi + PRE, n)
        tmp = np.zeros(n, np.int64)
        trace(dec, end if top == n else best[top], top, lo, tmp)
        seg[lo:hi] = tmp[lo:hi]
    tail = 200                                                       # the last steps of a stream are decided with little look-ahead by either
    e_one, e_seg = int((one[:n - tail] != msg[:n - tail]).sum()), int((seg[:n - tail] != msg[:n - tail]).sum())
    differ = int((one[:n - tail] != seg[:n - tail]).sum())
    assert 0 < e_one < 0.05 * n, e_one                               # the point lies at the waterfall: errors exist, the decoder works
    assert differ <= 8 and abs(e_one - e_seg) <= 8, (differ, e_one, e_seg)
