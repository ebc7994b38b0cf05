"""SURVEY 8e on one GPU: ONE stream decoded as a single segment and as 2..4 overlapping pieces on separate handles and
HIP streams (gr_dvbt_amd/multi.py: plan_cuts -> dvbt_rx_set_cut -> stitch_ts).  The stitched TS must equal the single
segment's TS byte for byte -- including the 8k QAM64 d_fi_start = 2 start one frame early
(demod_reference_signals_impl.cc:73-77), the 2244-byte fill of the byte de-interleaver
(convolutional_deinterleaver_impl.cc:64-65), the Viterbi restart and the descrambler's re-lock of every piece."""
import numpy as np
import pytest
import torch

import gr_dvbt_amd as g
from gr_dvbt_amd import multi

pytestmark = pytest.mark.gpu


def decode_pieces(const, cr, mode, iq, cuts, concurrent=True):
    """every piece on its own handle and stream; enqueue all, then collect"""
    dev = torch.from_numpy(iq.view(np.float32)).cuda()
    torch.cuda.synchronize()
    rxs, streams = [], []
    for cu in cuts:
        rx = g.Rx(const, cr, mode, max_samples=cu["end"] - cu["begin"])
        rx.set_cut(cu["sym_off"])
        st = torch.cuda.Stream()
        rx.enqueue_device(dev.data_ptr() + 8 * cu["begin"], cu["end"] - cu["begin"], st.cuda_stream)
        if not concurrent:
            rx.finish()
        rxs.append(rx); streams.append(st)
    out = []
    for rx in rxs:
        rep = rx.finish() if concurrent else rx.report
        out.append((rep, rx.tap(g.TAP_TS)))
    for rx in rxs:
        rx.close()
    return out


@pytest.mark.parametrize("const,cr,mode,nsf,parts_list", [
    (g.QAM64, g.C7_8, g.T8k, 12, (2, 3, 4)),
    (g.QAM16, g.C1_2, g.T2k, 9, (2, 4)),
    (g.QPSK, g.C7_8, g.T8k, 5, (3,)),
])
def test_stitched_pieces_equal_the_single_segment(po, const, cr, mode, nsf, parts_list):
    c = po.cfg(const, cr, mode)
    d = g.get_dims(const, cr, mode)
    iq = po.stream_slice(c, nsf, 21)
    rx = g.Rx(const, cr, mode, max_samples=len(iq))
    rep = rx.run(iq)
    one = rx.tap(g.TAP_TS)
    rx.close()
    assert rep.status in (0, 2) and rep.segment_offset == 0 and len(one) > 0      # 2: the lock ends with the signal (zero tail)
    sf_call = rep.first_call + rep.first_out_symbol
    # what was transmitted: the TS tap starts ts_first_packet RS words after the superframe start, and RS word w carries the
    # packet sent 11 words earlier (end-to-end delay of the Forney interleaver pair)
    pps = po.packets_per_superframe(c)
    p0 = (rep.first_out_symbol * (c.payload * c.m * c.k // c.n) // 8) // 204 + rep.ts_first_packet - 11
    sent = po.stream_ts(c, 0, nsf, 21)
    assert (one == sent[p0 * 188:p0 * 188 + len(one)]).all() and p0 < 2 * pps
    for parts in parts_list:
        cuts = multi.plan_cuts(d, len(iq), int(rep.segment_offset), sf_call, parts)
        assert len(cuts) == parts
        pieces = decode_pieces(const, cr, mode, iq, cuts)
        for (r, _), cu in zip(pieces, cuts):
            assert r.status in (0, 2) and r.stream_symbol_offset == cu["sym_off"]
        assert all(r.first_out_symbol == multi.PRE_SYMBOLS for r, _ in pieces[1:])
        st = multi.stitch_ts(pieces, d)
        assert len(st) == len(one) and (st == one).all(), (parts, len(st), len(one))


def test_cut_mode_piece_equals_the_oracle_cut_mode(po):
    """the checker of the cut mode itself: every tap of a continuation piece against o_rx_run_cut"""
    const, cr, mode = g.QAM64, g.C7_8, g.T8k
    c = po.cfg(const, cr, mode)
    d = g.get_dims(const, cr, mode)
    iq = po.stream_slice(c, 3, 4)
    cuts = multi.plan_cuts(d, len(iq), 0, 204, 2)
    cu = cuts[1]
    seg = iq[cu["begin"]:cu["end"]]
    ref = po.rx(c, seg, want=("vit", "rs", "ts"), sym_off=cu["sym_off"])
    rx = g.Rx(const, cr, mode, max_samples=len(seg))
    rx.set_cut(cu["sym_off"])
    rep = rx.run(seg)
    assert rep.first_out_symbol == ref["first_out_symbol"] and rep.ts_first_packet == ref["ts_first_packet"]
    assert rep.stream_rs_items == ref["stream_rs_items"]
    for tap, key in ((g.TAP_VITERBI, "vit"), (g.TAP_RS, "rs"), (g.TAP_TS, "ts")):
        a = rx.tap(tap)
        assert len(a) == len(ref[key]) > 0 and (a == ref[key]).all(), key
    rx.close()


def test_set_cut_rejects_partial_superframes():
    rx = g.Rx(g.QAM16, g.C1_2, g.T2k, max_samples=100000)
    with pytest.raises(g.DvbtError):
        rx.set_cut(100)
    rx.set_cut(272)
    rx.close()
