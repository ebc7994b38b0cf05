"""Segments in flight (bench.py --pipeline): several handles, each on its own HIP stream, decode at the same time and
share the machine -- the persistent 8k symbol kernel takes its symbols from an atomic counter and its workgroups start whenever the other
segments' Viterbi workgroups make room.  Every handle must deliver exactly what a handle that has the GPU to itself delivers -- and that is the
ORACLE's chain over the same samples (po.rx), so the test never compares the HIP path with itself alone."""
import numpy as np
import pytest
import torch

import gr_dvbt_amd as g

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("const,cr,mode,nsf", [(g.QAM64, g.C7_8, g.T8k, 6), (g.QAM16, g.C2_3, g.T8k, 4), (g.QAM16, g.C1_2, g.T2k, 8)])
def test_handles_in_flight_equal_a_handle_alone(po, const, cr, mode, nsf):
    c = po.cfg(const, cr, mode)
    ibits = c.payload * c.m * c.k // c.n
    # three different streams (so that a buffer mixed up between handles cannot go unnoticed) of the same configuration
    iqs = [po.tx(c, po.make_ts((272 * ibits * nsf) // (204 * 8), 100 + k), lead_in=500 + 700 * k, tail=3 * c.N) for k in range(3)]
    alone = []
    for iq in iqs:
        o = po.rx(c, iq, want=("vit", "ts"))                   # the reference bytes: the oracle's chain (clean loopback: bit-exact at every integer tap)
        rx = g.Rx(const, cr, mode, max_samples=len(iq))
        rep = rx.run(iq)
        assert (rep.n_symbols, rep.first_out_symbol) == (o["n_acquired"], o["first_out_symbol"])
        assert len(o["ts"]) == rep.n_ts_bytes > 0 and (rx.tap(g.TAP_TS) == o["ts"]).all() and (rx.tap(g.TAP_VITERBI) == o["vit"]).all()
        alone.append((rep.n_symbols, rep.first_out_symbol, int(rep.n_ts_bytes), o["vit"].copy(), o["ts"].copy()))
        rx.close()
    devs = [torch.from_numpy(iq.view(np.float32)).cuda() for iq in iqs]
    torch.cuda.synchronize()
    rxs = [g.Rx(const, cr, mode, max_samples=len(iq)) for iq in iqs]
    streams = [torch.cuda.Stream() for _ in iqs]
    for rnd in range(3):                                       # the handles are reused, as a receiver working through a stream does
        for rx, d, iq, st in zip(rxs, devs, iqs, streams):
            rx.enqueue_device(d.data_ptr(), len(iq), st.cuda_stream)
        for k, rx in enumerate(rxs):
            rep = rx.finish()
            n_sym, first, n_ts, vit, ts = alone[k]
            assert (rep.n_symbols, rep.first_out_symbol, int(rep.n_ts_bytes)) == (n_sym, first, n_ts), (rnd, k)
            assert (rx.tap(g.TAP_VITERBI) == vit).all() and (rx.tap(g.TAP_TS) == ts).all(), (rnd, k)
    for rx in rxs:
        rx.close()


def test_step_replayed_as_a_hip_graph(po):
    """dvbt_rx_params.launch_graph: the launch sequence of dvbt_rx_segment_enqueue_device captured once per (segment, length, stream, cut) and replayed as one graph
    launch -- the same bytes as launch by launch (the oracle's), for a second segment on the same handle too, and again after the first one's turn"""
    const, cr, mode = g.QAM64, g.C7_8, g.T8k
    c = po.cfg(const, cr, mode)
    ibits = c.payload * c.m * c.k // c.n
    iqs = [po.tx(c, po.make_ts((272 * ibits * 3) // (204 * 8), 300 + k), lead_in=400 + 300 * k, tail=3 * c.N) for k in range(2)]
    refs = [po.rx(c, iq, want=("ts",))["ts"] for iq in iqs]
    n = max(len(iq) for iq in iqs)
    devs = [torch.from_numpy(iq.view(np.float32)).cuda() for iq in iqs]
    torch.cuda.synchronize()
    rx = g.Rx(const, cr, mode, max_samples=n, launch_graph=1)
    st = torch.cuda.Stream()
    for rnd in range(3):
        for k in (0, 1):
            rx.enqueue_device(devs[k].data_ptr(), len(iqs[k]), st.cuda_stream)
            rep = rx.finish()
            ts = rx.tap(g.TAP_TS)
            assert rep.n_ts_bytes == len(refs[k]) > 0 and (ts == refs[k]).all(), (rnd, k)
    rx.close()


def test_front_end_on_a_priority_stream(po):
    """dvbt_rx_params.front_priority: the front end of a segment on the handle's own high-priority stream, joined in front of the Viterbi decoder -- two handles in
    flight, the oracle's bytes from both, round after round"""
    const, cr, mode = g.QAM64, g.C7_8, g.T8k
    c = po.cfg(const, cr, mode)
    ibits = c.payload * c.m * c.k // c.n
    iqs = [po.tx(c, po.make_ts((272 * ibits * 3) // (204 * 8), 500 + k), lead_in=300 + 500 * k, tail=3 * c.N) for k in range(2)]
    refs = [po.rx(c, iq, want=("ts",))["ts"] for iq in iqs]
    devs = [torch.from_numpy(iq.view(np.float32)).cuda() for iq in iqs]
    torch.cuda.synchronize()
    rxs = [g.Rx(const, cr, mode, max_samples=len(iq), front_priority=1) for iq in iqs]
    streams = [torch.cuda.Stream() for _ in iqs]
    for rnd in range(3):
        for rx, d, iq, st in zip(rxs, devs, iqs, streams):
            rx.enqueue_device(d.data_ptr(), len(iq), st.cuda_stream)
        for k, rx in enumerate(rxs):
            rep = rx.finish()
            assert rep.n_ts_bytes == len(refs[k]) > 0 and (rx.tap(g.TAP_TS) == refs[k]).all(), (rnd, k)
    for rx in rxs:
        rx.close()
