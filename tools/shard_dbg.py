"""A sharded stream through a dropout, traced: each rank's chunks (first packet label, packets), where the ranks' TS leaves the single chain's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main(hole_symbol=272 * 7 + 100, seg_sf=4, world=2, nsf=15, verbose=1):
    import torch  # noqa: F401
    from oracle import pyoracle as po
    import gr_dvbt_amd as g
    const, cr, mode = g.QAM16, g.C1_2, g.T2k
    c = po.cfg(const, cr, mode)
    L = c.N + c.cp
    iq = po.stream_slice(c, nsf, 9).copy()
    a = po.STREAM_LEAD_IN + hole_symbol * L
    iq[a:a + 30 * L] = 0
    rx = g.Rx(const, cr, mode, max_samples=len(iq)); rx.run(iq); ref = rx.tap(g.TAP_TS).copy()
    print("single chain periods", rx.lock_periods(), len(ref) // 188, "packets")
    rx.close()
    ranks = [g.RxStream(const, cr, mode, segment_superframes=seg_sf, rank=r, world=world) for r in range(world)]
    chunks = []
    step = 64 * L
    for a in range(0, len(iq), step):
        for st in ranks:
            st.push(iq[a:a + step])
        for r, st in enumerate(ranks):
            chunks += [(fp, r, b) for fp, b in st.pull_chunks()]
    for r, st in enumerate(ranks):
        st.finish()
        chunks += [(fp, r, b) for fp, b in st.pull_chunks()]
    stat = [st.info().status for st in ranks]
    for r, st in enumerate(ranks):
        if verbose:
            print("--- rank", r, "status", st.info().status); print(st.trace())
        st.close()
    chunks.sort(key=lambda t: t[0])
    at, pos = None, 0
    refp = ref.reshape(-1, 188)
    index = {}
    for k, p in enumerate(refp):
        index.setdefault(bytes(p), k)
    for fp, r, b in chunks:
        n = len(b) // 188
        where = index.get(bytes(b[:188]), -1)
        same = pos + n <= len(refp) and (b.reshape(-1, 188) == refp[pos:pos + n]).all()
        flag = "" if same else "  <-- differs from the single chain at this place (its first packet is the chain's packet %d)" % where
        if at is not None and fp != at or not same:
            print("chunk label %d rank %d packets %d (running count %d, label gap %s)%s" % (fp, r, n, pos, None if at is None else fp - at, flag))
        at = fp + n; pos += n
    ts = np.concatenate([b for _, _, b in chunks])
    print("hole", hole_symbol, "pieces of", seg_sf, "world", world, "status", stat, "total", pos, "packets; single chain", len(refp), "equal", len(ts) == len(ref) and bool((ts == ref).all()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "scan":
        for hs in (272 * 6 + 100, 272 * 5 + 150, 272 * 7 + 40, 272 * 6 + 200, 272 * 9 + 60, 272 * 10 + 130, 272 * 7 + 100, 272 * 8 + 100):
            main(hs, verbose=0)
    else:
        main(*[int(x) for x in sys.argv[1:]])
