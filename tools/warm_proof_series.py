"""The chunk decoders' warm-up where the code copes with the channel, on many chunk starts (CPU, the oracle's decoder; tools/hier_warmup.py holds the collapsed regimes): for every block
boundary of long streams at a pre-Viterbi bit error rate of 1 % and 2 % (rate 7/8, the worst puncturing) -- does a decoder started 72 windows early decode the streaming decoder's bytes,
and is its metric vector at the chunk's first window the streaming decoder's (what dvbt_rx_params.viterbi_verify compares: a chunk that is flagged is decoded again for nothing)?
python tools/warm_proof_series.py [streams per case [bit error rates, comma separated]] > profiles/rNN_warm_proof_series.json"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402

L = po.lib()
L.o_viterbi_decode_snap.restype = C.c_size_t
L.o_viterbi_decode_snap.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
W = 72


def series(c, vin):
    blk_in, blk_out = 768 * c.n // c.m, 768 * c.k // 8
    nb = len(vin) // blk_in
    vin = np.ascontiguousarray(vin[:nb * blk_in])
    starts = np.arange(1, nb - 4)
    at = (starts * blk_out + W).astype(np.int64)
    full = np.zeros(nb * blk_out + 64, np.uint8)
    snaps = np.zeros((len(at), 64), np.uint8)
    nfull = L.o_viterbi_decode_snap(C.byref(c), vin.ctypes.data, len(vin), full.ctypes.data, at.ctypes.data, len(at), snaps.ctypes.data)
    one = np.array([W], np.int64)
    differ = flagged = broken = 0
    for i, j in enumerate(starts):
        sub = vin[j * blk_in:(j + 3) * blk_in]
        o = np.zeros(3 * blk_out + 64, np.uint8)
        s = np.zeros(64, np.uint8)
        n = L.o_viterbi_decode_snap(C.byref(c), sub.ctypes.data, len(sub), o.ctypes.data, one.ctypes.data, 1, s.ctypes.data)
        m = min(n, nfull - j * blk_out)
        same_bytes = bool((o[W - 1:m] == full[j * blk_out + W - 1:j * blk_out + m]).all())
        same_state = bool((s == snaps[i]).all())
        differ += not same_bytes
        flagged += not same_state
        broken += same_state and not same_bytes                         # must stay 0: equal vectors, different bytes
    return len(starts), differ, flagged, broken


def main():
    nstreams = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    bers = tuple(float(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (0.01, 0.02)
    rows = []
    for const, label in ((po.QAM64, "QAM64 7/8"), (po.QPSK, "QPSK 7/8")):
        c = po.cfg(const, po.C7_8, po.T2k)
        ibits = c.payload * c.m * c.k // c.n
        for ber in bers:
            tot = [0, 0, 0, 0]
            for k in range(nstreams):
                iq = po.tx(c, po.make_ts((272 * ibits * 40) // (204 * 8), 100 + k), lead_in=500, tail=3 * c.N)
                vin = po.rx(c, iq, want=("bitdeint",))["bitdeint"].reshape(-1).copy()
                rng = np.random.RandomState(1000 + k)
                for b in range(c.m):
                    vin ^= (rng.rand(len(vin)) < ber).astype(np.uint8) << b
                r = series(c, vin)
                tot = [a + b for a, b in zip(tot, r)]
                print(label, ber, k, r, file=sys.stderr, flush=True)
            rows.append({"stream": label, "bit_error_rate": ber, "warm_up_windows": W, "chunk_starts": tot[0], "decode_differently": tot[1],
                         "metric_vector_differs_at_the_chunk_start": tot[2], "equal_vector_but_different_bytes": tot[3]})
    print(json.dumps({"tool": "tools/warm_proof_series.py", "decoder": "oracle/o_viterbi.c (pinned to the reference's kernels)", "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
