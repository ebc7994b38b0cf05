"""Soft-decision mode (dvbt_rx_params.soft_decision = 1; SURVEY 8f row 4, second half; gr-dvbt's TODO.txt:25-26).  The reference decodes hard decisions only:
there is no oracle for this mode and no parity claim.  It is validated against the hard path on identical samples: the same TS, byte for byte, on clean
loopbacks (and on noisy ones as long as the hard path still decodes everything), and a working decoder where the hard path has collapsed -- at 1 dB and at
2 dB below the SNR at which the hard path's packet error rate passes 50 %."""
import numpy as np
import pytest

import gr_dvbt_amd as g

pytestmark = pytest.mark.gpu


def both(po, const, cr, mode, nsf, snr=None, echoes=(), seed=9, rx_snr=None):
    c = po.cfg(const, cr, mode)
    clean = po.stream_slice(c, nsf, seed)
    iq = po.channel(clean, c.N, echoes=echoes, snr_db=snr, seed=5) if (snr is not None or echoes) else clean
    sent = {bytes(p) for p in po.stream_ts(c, 0, nsf, seed).reshape(-1, 188)}
    out = []
    for soft in (0, 1):
        rx = g.Rx(const, cr, mode, max_samples=len(iq), snr_db=rx_snr if rx_snr is not None else 30.0 if snr is None else snr, soft_decision=soft)
        rep = rx.run(iq)
        ts = rx.tap(g.TAP_TS).copy()
        pk = ts.reshape(-1, 188)
        per = 1.0 if len(pk) == 0 else 1.0 - sum(1 for p in pk if bytes(p) in sent) / len(pk)
        out.append({"ts": ts, "rep": rep, "per": per, "fail": rep.rs_fail_words, "corr": rep.rs_corrected_symbols, "periods": rep.n_lock_periods})
        rx.close()
    return out


@pytest.mark.parametrize("const,cr,mode,nsf", [(g.QAM16, g.C1_2, g.T2k, 3), (g.QAM64, g.C7_8, g.T8k, 2), (g.QPSK, g.C2_3, g.T8k, 2), (g.QAM64, g.C3_4, g.T2k, 3)])
def test_clean_loopback_equals_the_hard_path(po, const, cr, mode, nsf):
    hard, soft = both(po, const, cr, mode, nsf)
    assert len(soft["ts"]) == len(hard["ts"]) > 0 and (soft["ts"] == hard["ts"]).all()
    assert soft["fail"] == hard["fail"] == 11 and soft["corr"] == 0


@pytest.mark.parametrize("const,cr,mode,nsf,snr_ok,snr_hard_dead", [
    (g.QAM16, g.C1_2, g.T2k, 4, 12.0, 10.0),          # hard: 787 corrected symbols at 12 dB, PER 0.91 at 10 dB
    (g.QAM64, g.C7_8, g.T8k, 2, 22.0, 20.0),          # hard: 12,558 corrected at 22 dB, PER 1.0 at 20 dB
    (g.QAM64, g.C2_3, g.T2k, 4, 20.0, 17.0),
])
def test_noise_soft_decodes_where_hard_has_collapsed(po, const, cr, mode, nsf, snr_ok, snr_hard_dead):
    hard, soft = both(po, const, cr, mode, nsf, snr=snr_ok)
    # both still deliver every packet: identical TS; the soft decoder leaves the RS decoder (much) less to do
    assert hard["per"] < 0.02 and (soft["ts"] == hard["ts"]).all()
    assert hard["corr"] > 100 and soft["corr"] <= hard["corr"] // 10
    hard, soft = both(po, const, cr, mode, nsf, snr=snr_hard_dead)
    assert hard["per"] > 0.5, "the point is meant to lie below the hard path's waterfall"
    assert soft["periods"] == hard["periods"] == 1
    assert soft["per"] < 0.01 and soft["fail"] <= 11 + 2


def test_frequency_selective_channel(po):
    """an echo at -10 dB: carriers in the fades carry less weight (channel state information in the LLRs)"""
    hard, soft = both(po, g.QAM16, g.C3_4, g.T2k, 4, snr=16.0, echoes=((19, 0.3),))
    assert soft["periods"] == hard["periods"] and len(soft["ts"]) == len(hard["ts"]) > 0
    assert soft["corr"] < hard["corr"] and soft["per"] <= hard["per"]


def test_deep_echo_below_the_hard_waterfall(po):
    """an echo at -6 dB inside the guard interval (notches 9.5 dB below the peaks), 16 dB of noise: the hard path has collapsed (PER 0.71 over 8 superframes,
    tools/soft_gain.py echo), the soft path with its channel-state weights delivers every packet (3 dB of gain on this channel against 2.3 dB on a flat one).
    ofdm_sym_acquisition's snr parameter at 10 dB: at the demo flowgraphs' 30 dB the reference's peak detector drops the lock on this channel."""
    hard, soft = both(po, g.QAM16, g.C3_4, g.T2k, 8, snr=16.0, echoes=((20, 0.5 + 0j),), rx_snr=10.0)
    assert soft["periods"] == hard["periods"] == 1
    assert hard["per"] > 0.3 and soft["per"] < 0.01


@pytest.mark.parametrize("const,cr,mode,nsf,seg_sf,call", [(g.QAM16, g.C1_2, g.T2k, 9, 2, 4 * 2112), (g.QAM64, g.C7_8, g.T8k, 6, 1, 33 * 8448 + 5)])
def test_soft_mode_through_the_streaming_entry(po, const, cr, mode, nsf, seg_sf, call):
    """dvbt_rx_stream_* with soft_decision = 1: pieces that continue a cut stream (dvbt_rx_set_cut: sizes in stream coordinates) go through the soft decoder
    too; on a clean stream the TS pulled is the hard chain's over the whole stream"""
    c = po.cfg(const, cr, mode)
    iq = po.stream_slice(c, nsf, 9)
    rx = g.Rx(const, cr, mode, max_samples=len(iq))
    rx.run(iq)
    ref = rx.tap(g.TAP_TS).copy()
    rx.close()
    st = g.RxStream(const, cr, mode, segment_superframes=seg_sf, soft_decision=1)
    out = []
    for pos in range(0, len(iq), call):
        st.push(iq[pos:pos + call])
        out.append(st.pull())
    st.finish()
    out.append(st.pull())
    info = st.info()
    st.close()
    ts = np.concatenate(out)
    assert info.status & ~2 == 0 and len(ts) == len(ref) > 0 and (ts == ref).all()
