"""TPS auto-detection (the reference author's TODO.txt:25-28): the report carries the transmission parameters
the stream itself signals -- the TPS word of a BCH-valid frame (reference_signals_impl.cc:385-425) decoded per the layout of
format_tps_data (:883-916) -- and a handle configured differently is told so."""
import ctypes as C
import numpy as np
import pytest

import gr_dvbt_amd as g

pytestmark = pytest.mark.gpu

# the five BASELINE configurations (1 and 2 share their parameters) + one more rate
CONFIGS = [(g.QAM16, g.C1_2, g.T2k), (g.QAM64, g.C7_8, g.T8k), (g.QPSK, g.C7_8, g.T8k), (g.QAM64, g.C3_4, g.T2k)]


@pytest.mark.parametrize("const,cr,mode", CONFIGS)
def test_report_carries_the_signalled_parameters(po, const, cr, mode):
    c = po.cfg(const, cr, mode)
    iq = po.stream_slice(c, 2, 3)
    rx = g.Rx(const, cr, mode, max_samples=len(iq))
    rep = rx.run(iq)
    rx.close()
    assert rep.tps_valid == 1 and rep.tps_mismatch == 0 and not rep.status & 16
    assert (rep.tps_constellation, rep.tps_hierarchy, rep.tps_code_rate_hp, rep.tps_code_rate_lp) == (const, g.NH, cr, cr)
    assert (rep.tps_guard_interval, rep.tps_transmission_mode, rep.tps_cell_id) == (g.G1_32, mode, 0)
    assert rep.tps_length_indicator == 0x17                       # no cell id (:898-901)
    # the raw word against the transmitter's (o_tps_format, pinned by the Appendix F known-answer test): bits s17-s22, s25-s53
    wk = np.zeros(c.Kmax + 1, np.int8)
    po.lib().o_prbs_wk(C.byref(c), wk.ctypes.data_as(C.c_void_p))
    tps = np.zeros(68, np.uint8)
    po.lib().o_tps_format(C.byref(c), 1, wk.ctypes.data_as(C.c_void_p), tps.ctypes.data_as(C.c_void_p))
    want = sum(int(tps[i]) << i for i in list(range(17, 23)) + list(range(25, 54)))
    assert rep.tps_bits == want


def test_a_wrongly_configured_handle_is_told(po):
    """8k QAM64 7/8 on the air, the handle set up for 8k QAM16 2/3: the front end is the same, so the TPS is received and decoded,
    and the report names what differs (constellation and code rate)."""
    c = po.cfg(g.QAM64, g.C7_8, g.T8k)
    iq = po.stream_slice(c, 2, 5)
    rx = g.Rx(g.QAM16, g.C2_3, g.T8k, max_samples=len(iq))
    rep = rx.run(iq)
    rx.close()
    assert rep.tps_valid == 1 and rep.status & 16
    assert rep.tps_mismatch == 1 | 4
    assert (rep.tps_constellation, rep.tps_code_rate_hp, rep.tps_transmission_mode) == (g.QAM64, g.C7_8, g.T8k)
    from gr_dvbt_amd import multi
    with pytest.raises(ValueError):                               # such a piece is not stitched
        multi.stitch_plan([multi.piece_meta(rep)], g.get_dims(g.QAM16, g.C2_3, g.T8k))


def test_no_signal_no_tps(po):
    rng = np.random.RandomState(1)
    iq = (1e-3 * (rng.randn(200000) + 1j * rng.randn(200000))).astype(np.complex64)
    rx = g.Rx(g.QAM16, g.C1_2, g.T2k, max_samples=len(iq))
    rep = rx.run(iq)
    rx.close()
    assert rep.tps_valid == 0 and rep.tps_bits == 0 and rep.tps_mismatch == 0
