"""oracle/o_soft.c is a MODEL of the product's soft-decision kernels (the reference has no soft path); the GPU tests require the kernels to equal it.  That only means
something if the model is itself a correct decoder: here it decodes what an encoder written from ETSI EN 300 744 4.3.3 (mother code 171 / 133 octal, the puncturing
patterns of table 4) produces -- exactly on a clean stream for every code rate, chunk boundaries and traceback segments included, and with a plausible error rate
under noise (better than hard decisions on the same samples)."""
import ctypes as C
import numpy as np
import pytest

PUNCT = {0: [1, 1], 1: [1, 1, 0, 1], 2: [1, 1, 0, 1, 1, 0], 3: [1, 1, 0, 1, 1, 0, 0, 1, 1, 0], 4: [1, 1, 0, 1, 0, 1, 0, 1, 1, 0, 0, 1, 1, 0]}
NTB = {0: 5, 1: 9, 2: 10, 3: 15, 4: 24}


def encode(bits):
    u = np.concatenate([np.zeros(6, np.uint8), bits.astype(np.uint8)])
    n = len(bits)
    tap = lambda d: u[6 - d:6 - d + n]
    x = tap(0) ^ tap(1) ^ tap(2) ^ tap(3) ^ tap(6)          # 171 octal
    y = tap(0) ^ tap(2) ^ tap(3) ^ tap(5) ^ tap(6)          # 133 octal
    out = np.empty(2 * n, np.uint8); out[0::2] = x; out[1::2] = y
    return out


@pytest.mark.parametrize("cr", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("sigma", [0.0, 1.0])
def test_model_decoder_decodes(po, cr, sigma):
    rng = np.random.RandomState(40 + cr)
    nbytes = 6000
    data = rng.randint(0, 256, nbytes).astype(np.uint8)
    bits = np.unpackbits(data)
    coded = encode(bits)
    mask = np.tile(np.array(PUNCT[cr], bool), len(coded) // len(PUNCT[cr]) + 1)[:len(coded)]
    tx = coded[mask]
    sigma *= {0: 5.0, 1: 4.0, 2: 3.3, 3: 2.8, 4: 2.4}[cr]               # near each rate's waterfall
    soft = np.clip(np.rint(8.0 * (1 - 2.0 * tx) + sigma * rng.randn(len(tx))), -31, 31).astype(np.int8)
    c = po.cfg(po.QAM16, cr, po.T2k)
    L = po.lib()
    L.o_soft_plan.argtypes = [C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]
    L.o_soft_viterbi.restype = C.c_longlong
    L.o_soft_viterbi.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_void_p]
    total_steps = len(bits)
    out = np.zeros(nbytes + 64, np.uint8)
    for B in (64, 137, 304):                                           # chunk sizes: the smallest, an odd one, the largest (several chunks, eight traceback segments each)
        look = max(8 * NTB[cr], 128)
        nsteps = (256 + 8 * B + look + 47) // 48 * 48
        n = L.o_soft_viterbi(C.byref(c), soft.ctypes.data, len(soft), total_steps, B, nsteps, out.ctypes.data)
        assert n == nbytes - NTB[cr]
        good = n - 40                                                  # the stream's last bytes are decided with little look-ahead
        err = np.unpackbits(out[:good] ^ data[:good]).sum()
        if sigma == 0.0:
            assert err == 0, (cr, B, err)
        else:
            hard = (soft < 0).astype(np.uint8)                         # hard decisions of the same samples, decoded by the same model with unit soft values
            hs = np.where(hard == 1, -8, 8).astype(np.int8)
            out2 = np.zeros_like(out)
            L.o_soft_viterbi(C.byref(c), hs.ctypes.data, len(hs), total_steps, B, nsteps, out2.ctypes.data)
            err_hard = np.unpackbits(out2[:good] ^ data[:good]).sum()
            assert err <= err_hard and err < 0.02 * 8 * good, (cr, B, err, err_hard)
