"""the stock rational_resampler_ccc(64, 70) + multiply_const in front of the RX path.

Third-party stage (gr-filter / gr-fft, absent and unpinned): parity unpinned; the oracle restates the published
GNU Radio 3.7 design (oracle/o_resample.c), these tests pin the restatement to the closed-form facts of that design
and the HIP kernel to the oracle within a float tolerance (VOLK's summation order is not defined either)."""
import ctypes as C
import numpy as np
import pytest


def test_design_facts(po):
    t, ri, rd = po.resampler_taps(64, 70)
    assert (ri, rd) == (32, 35)                       # reduced by the gcd when no taps are given
    # compute_ntaps: (beta/0.1102 + 8.7) * fs / (22 * width), width = (32/35) * 0.1, fs = 32, made odd
    assert len(t) == int((7.0 / 0.1102 + 8.7) * 32 / (22.0 * (32 / 35) * 0.1)) + 1 == 1149
    assert np.allclose(t, t[::-1], atol=1e-9)         # linear phase
    assert abs(float(t.astype(np.float64).sum()) - 32.0) < 1e-4   # DC gain = interpolation
    H = np.abs(np.fft.rfft(t.astype(np.float64), 1 << 17)) / 32.0
    f = np.arange(len(H)) / (1 << 17) * 32                # in units of the input rate
    assert H[f < 0.36].min() > 0.99                   # pass band up to (32/35) * 0.4
    assert abs(H[np.argmin(np.abs(f - (32 / 35) * 0.45))] - 0.5) < 0.01   # -6 dB in the middle of the transition band
    assert H[f > 0.48].max() < 10 ** (-70 / 20)       # Kaiser beta 7: ~72 dB stop band


def test_tone_and_rate(po):
    fs = 10e6
    x = np.exp(2j * np.pi * 1.25e6 * np.arange(70000) / fs).astype(np.complex64)
    y = po.resample(x, 64, 70, 0.5)
    assert len(y) == 64000
    ss = y[2000:60000]
    assert np.abs(np.abs(ss) - 0.5).max() < 2e-4      # unity pass-band gain, then the scale
    ph = np.angle(ss[1:] * np.conj(ss[:-1])).mean()
    assert abs(ph / (2 * np.pi) * fs * 64 / 70 - 1.25e6) < 5.0


@pytest.fixture(scope="module")
def g():
    import gr_dvbt_amd
    assert gr_dvbt_amd.device_count() > 0
    return gr_dvbt_amd


@pytest.mark.gpu
def test_resampler_block_streaming(po, g):
    rng = np.random.RandomState(3)
    x = (rng.randn(200000) + 1j * rng.randn(200000)).astype(np.complex64)
    want = po.resample(x, 64, 70, 0.00055242272)
    b = g.Block("resampler", 64, 70, 0.00055242272)
    assert b.forecast(3200) == 3200 * 35 // 32 + 35
    outs, pos = [], 0
    for nin in (1, 35, 1000, 70001, 4096, 200000):                 # ragged work() calls; state carried across them
        nin = min(nin, len(x) - pos)
        if nin <= 0:
            break
        o = np.zeros(nin + 8, np.complex64)
        r, cons, _ = b.work(len(o), nin, x[pos:pos + nin].copy(), o)
        assert cons == nin or r == len(o)
        outs.append(o[:r]); pos += cons
    got = np.concatenate(outs)
    assert pos == len(x) and len(got) == len(want)
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max() * 36 ** 0.5
    b.close()


@pytest.mark.gpu
def test_chain_from_file_rate(po, g):
    """10 Msps 'file format' in -> TS out: resample + scale on the device, then the whole chain; TS identical to the
    oracle chain fed with the oracle-resampled stream."""
    c = po.cfg(1, 0, 0)                                            # 2k QAM16 1/2
    ibits = c.payload * c.m * c.k // c.n
    ts = po.make_ts((272 * ibits * 3) // (204 * 8), 77)
    iq = po.tx(c, ts, scale=1.0 / (10 * np.sqrt(2048.0)), lead_in=1000, tail=3 * c.N)   # TX at 64/7 Msps before its own scaling
    file_rate = po.resample(iq, 70, 64, 1.0)                        # what dvbt_tx_demo writes: the 10 Msps stream
    scale = 0.0022097087
    at_tap = po.resample(file_rate, 64, 70, scale)
    o = po.rx(c, at_tap, want=("ts", "rs"))
    assert o["ts"].size > 100000
    rx = g.Rx(1, 0, 0, max_samples=len(file_rate), resample=(64, 70), front_scale=scale)
    rep = rx.run(file_rate)
    assert rep.status == 0 or rep.status == 2
    got = rx.tap(g.TAP_TS)
    assert got.size == o["ts"].size and (got == o["ts"]).all()
    rx.close()
