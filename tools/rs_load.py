"""The RS(204,188) kernel under error load: time of dvbt_reed_solomon_dec_work_device (deint_rs_kernel, standalone input) over W codewords
of which a fraction f carries e symbol errors each; output checked against the oracle's decoder on a sample.  One JSON line per (f, e).
`python tools/rs_load.py [words]` on the GPU box."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def make_words(po, nwords, seed=3):
    L = po.lib()
    rs = po.RS()
    L.o_rs_init(C.byref(rs))
    rng = np.random.RandomState(seed)
    base = 4096                                               # distinct codewords, tiled (encoding on the CPU is the slow part)
    data = np.zeros((base, 255), np.uint8)
    data[:, 51:239] = rng.randint(0, 256, size=(base, 188))
    for i in range(base):
        L.o_rs_encode(C.byref(rs), data[i].ctypes.data_as(C.c_void_p), data[i, 239:].ctypes.data_as(C.c_void_p))
    reps = (nwords + base - 1) // base
    return np.tile(data[:, 51:], (reps, 1))[:nwords].copy(), rs


def corrupt(words, frac, nerr, seed=7):
    rng = np.random.RandomState(seed)
    w = words.copy()
    n = len(w)
    bad = np.flatnonzero(rng.rand(n) < frac)
    for e in range(nerr):
        pos = (rng.randint(0, 204, size=len(bad)) + 17 * e) % 204 if nerr > 1 else rng.randint(0, 204, size=len(bad))
        w[bad, pos] ^= rng.randint(1, 256, size=len(bad)).astype(np.uint8)
    return w, bad


def run(po, g, nwords=344064, cases=None, iters=20):
    import torch
    words, rs = make_words(po, nwords)
    blk = g.Block("reed_solomon_dec", 2, 8, 0x11d, 255, 239, 8, 51, 8, 0)
    out = torch.empty(nwords * 188, dtype=torch.uint8, device="cuda")
    rows = []
    L = po.lib()
    for frac, nerr in (cases or [(0.0, 0), (0.01, 1), (0.1, 4), (0.5, 4), (1.0, 1), (1.0, 4), (1.0, 8), (1.0, 9)]):
        w, bad = corrupt(words, frac, nerr)
        d_in = torch.from_numpy(w.reshape(-1)).cuda()
        s = torch.cuda.current_stream()
        for _ in range(2):
            blk.work_device(nwords // 8, nwords // 8, d_in.data_ptr(), out.data_ptr(), (), s.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(iters):
            blk.work_device(nwords // 8, nwords // 8, d_in.data_ptr(), out.data_ptr(), (), s.cuda_stream)
        e1.record(s)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        got = out.cpu().numpy().reshape(nwords, 188)
        # check a sample against the oracle's decoder (correct mode)
        idx = np.unique(np.concatenate([bad[:200], np.arange(0, nwords, max(1, nwords // 200))])).astype(np.int64)
        ok = True
        for i in idx:
            cw = np.zeros(255, np.uint8)
            cw[51:] = w[i]
            L.o_rs_decode(C.byref(rs), cw.ctypes.data_as(C.c_void_p), 0)
            ok = ok and bool((cw[51:239] == got[i]).all())
        rows.append({"words": nwords, "bad_fraction": frac, "errors_per_bad_word": nerr, "ms": round(ms, 4),
                     "Mwords_per_s": round(nwords / ms / 1e3, 1), "GB_per_s_in_plus_out": round(nwords * 392 / ms / 1e6, 1), "sample_equals_oracle": ok})
    blk.close()
    return rows


if __name__ == "__main__":
    import torch  # noqa: F401
    from oracle import pyoracle as po
    import gr_dvbt_amd as g
    for r in run(po, g, int(sys.argv[1]) if len(sys.argv) > 1 else 344064):
        print(json.dumps(r), flush=True)
