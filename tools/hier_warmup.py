"""Why the hierarchical modes run ONE Viterbi decoder from the stream's start (dvbt_hip.hip "hierarchical modes"): a CPU experiment with the oracle's decoder.
The chunked decoder of the product starts every chunk 72 windows (576 trellis steps) early from all-zero metrics; it equals the streaming decoder where all survivors have merged
inside that warm-up.  Here a fresh decoder (oracle/o_viterbi.c = lib/d_viterbi.c's kernels in the block's calling pattern) is started at random block boundaries of the decoder's
input stream and its output is compared with the streaming decoder's from W bytes (= windows) behind its start, for several W: how many starts differ, how many bytes, and how far
behind the start the last differing byte lies.  Non-hierarchical streams are the control.  Runs on the CPU (no GPU): python tools/hier_warmup.py > profiles/rNN_hier_warmup.json"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402

L = po.lib()
L.o_viterbi_decode_n.restype = C.c_size_t
L.o_viterbi_decode_n.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]


def dec(c, inp):
    out = np.zeros(len(inp) * c.m * c.k // (8 * c.n) + 64, np.uint8)
    n = L.o_viterbi_decode_n(C.byref(c), inp.ctypes.data, len(inp), out.ctypes.data)
    return out[:n]


def experiment(c, vin, warms, nstarts, seed):
    blk_in, blk_out = 768 * c.n // c.m, 768 * c.k // 8
    nb = len(vin) // blk_in
    vin = np.ascontiguousarray(vin[:nb * blk_in])
    full = dec(c, vin)
    rng = np.random.RandomState(seed)
    starts = [int(rng.randint(1, nb - 12)) for _ in range(nstarts)]
    res = {}
    for W in warms:
        bad, nbytes, worst = 0, 0, 0
        for j in starts:
            o = dec(c, vin[j * blk_in:(j + 12) * blk_in])           # its output byte i is the stream's byte j * blk_out + i
            ref = full[j * blk_out:j * blk_out + len(o)]
            n = min(len(o), len(ref))
            d = np.flatnonzero(o[W:n] != ref[W:n])
            if len(d):
                bad += 1; nbytes += len(d); worst = max(worst, int(d[-1]) + W)
        res[str(W)] = {"starts_that_differ": bad, "bytes_that_differ": nbytes, "last_differing_byte_behind_the_start": worst}
    return {"blocks": nb, "starts": nstarts, "by_warm_up_windows": res}


def main():
    nstarts = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    rows = []
    for hier, name in ((0, "non-hierarchical"), (2, "hierarchical alpha 2, HP stream (output 0 of the bit de-interleaver)")):
        for const, cr, label in ((po.QAM64, po.C7_8, "2k QAM64 7/8"), (po.QAM64, po.C3_4, "2k QAM64 3/4"), (po.QAM64, po.C2_3, "2k QAM64 2/3"), (po.QAM16, po.C1_2, "2k QAM16 1/2")):
            c = po.cfg(const, cr, po.T2k, hierarchy=hier)
            ibits = c.payload * c.m * c.k // c.n
            iq = po.tx(c, po.make_ts((272 * ibits * 6) // (204 * 8), 5), lead_in=500, tail=3 * c.N)
            r = po.rx(c, iq, want=("bitdeint",))
            vin = r["bitdeint"][r["first_out_symbol"]:].reshape(-1)
            rows.append({"stream": f"{label}, {name}, clean loopback, 6 superframes", **experiment(c, vin, (48, 72, 96, 144, 288), nstarts, 7)})
            print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    print(json.dumps({"tool": "tools/hier_warmup.py", "decoder": "oracle/o_viterbi.c (restates lib/d_viterbi.c; pinned to the reference's own kernels by tests/test_oracle_ref_viterbi.py)",
                      "product_warm_up_windows": 72, "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
