"""Cutting one stream into pieces and stitching the decoded TS back (gr_dvbt_amd/multi.py) -- the host
logic and the cut-mode roundings, checked on the CPU with the oracle standing in for the chain (o_rx_run_cut).
tests/test_gpu_cut.py runs the same functions over the HIP chain."""
import numpy as np
import pytest

import gr_dvbt_amd as g
from gr_dvbt_amd import multi


def oracle_piece(po, c, iq, sym_off):
    r = po.rx(c, iq, want=("ts",), sym_off=sym_off)
    r["n_ts_bytes"] = len(r["ts"])
    r["status"] = 0 if r["first_out_symbol"] >= 0 else 4
    return r, r["ts"]


CASES = [
    (g.QAM16, g.C1_2, g.T2k, 7, (2, 3, 4)),       # d_fi_start = 3, 504 packets per superframe
    (g.QAM64, g.C7_8, g.T8k, 4, (2, 3)),          # d_fi_start = 2 (one frame early), 5292 packets per superframe (odd multiple of 4)
    (g.QPSK, g.C2_3, g.T2k, 6, (3,)),             # decoded bits per symbol not a whole number of bytes... per superframe they are
]


@pytest.mark.parametrize("const,cr,mode,nsf,parts_list", CASES)
def test_stitched_pieces_equal_the_single_chain(po, const, cr, mode, nsf, parts_list):
    c = po.cfg(const, cr, mode)
    d = g.get_dims(const, cr, mode)
    iq = po.stream_slice(c, nsf, 11)
    one = po.rx(c, iq, want=("ts",))
    assert one["first_out_symbol"] == (204 if (const == g.QAM64 and mode == g.T8k) else 272)
    for parts in parts_list:
        cuts = multi.plan_cuts(d, len(iq), 0, one["first_out_symbol"], parts)
        assert len(cuts) == parts and cuts[0]["begin"] == 0 and cuts[-1]["end"] == len(iq)
        pieces = [oracle_piece(po, c, iq[cu["begin"]:cu["end"]], cu["sym_off"]) for cu in cuts]
        for (r, _), cu in zip(pieces[1:], cuts[1:]):
            assert r["first_out_symbol"] == multi.PRE_SYMBOLS          # the pre-roll is exactly what the hunt consumes
        st = multi.stitch_ts(pieces, d)
        assert len(st) == len(one["ts"]) > 0 and (st == one["ts"]).all(), (parts, len(st), len(one["ts"]))


def test_stitch_refuses_a_short_post_roll(po):
    c = po.cfg(g.QAM16, g.C1_2, g.T2k)
    d = g.get_dims(g.QAM16, g.C1_2, g.T2k)
    iq = po.stream_slice(c, 5, 3)
    one = po.rx(c, iq, want=("ts",))
    cuts = multi.plan_cuts(d, len(iq), 0, one["first_out_symbol"], 2, post=4)
    pieces = [oracle_piece(po, c, iq[cu["begin"]:cu["end"]], cu["sym_off"]) for cu in cuts]
    with pytest.raises(ValueError, match="post-roll"):
        multi.stitch_ts(pieces, d)


def test_plan_cuts_gives_fewer_pieces_than_superframes_allow():
    d = g.get_dims(g.QAM64, g.C7_8, g.T8k)
    L = d.fft_length + d.cp_length
    cuts = multi.plan_cuts(d, 1000 + 3 * 272 * L + 3 * d.fft_length, 0, 204, 8)
    assert [c["count"] for c in cuts] == [1, 1] and cuts[1]["sym_off"] == 272
    assert cuts[1]["begin"] == (204 + 272 - multi.PRE_SYMBOLS) * L


def test_stream_slices_are_parts_of_one_stream(po):
    """bench.py: every rank generates only its piece of THE stream (oracle/pyoracle.py::stream_slice)."""
    c = po.cfg(g.QAM64, g.C7_8, g.T8k)
    L = c.N + c.cp
    full = po.stream_slice(c, 3, 5)
    assert len(full) == po.stream_len(c, 3)
    for b, e in ((1000 + (204 + 272 - 76) * L, None), (1000 + 272 * L + 3 * L, 1000 + 2 * 272 * L)):
        sl = po.stream_slice(c, 3, 5, b, e)
        assert (sl == full[b:e]).all()
