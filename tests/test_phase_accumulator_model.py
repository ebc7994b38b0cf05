"""Why the A1 / A2 taps of the HIP path and of the oracle differ by up to ~5e-4 of the peak under a fractional carrier offset (and not at all
without one): the reference keeps the derotation phase in a float and adds the (double) increment once per sample
(ofdm_sym_acquisition_impl.cc:285-309: `d_phase += d_phaseinc`, wrapped to [-pi, pi]).  While the accumulator stays inside one binade every step
rounds the same way, i.e. the phase advances by the increment rounded to that binade's grid: a rate error of up to half an ulp per sample that
changes from binade to binade.  Over the 8448 samples of an 8k symbol that is a wander of a few 1e-4 rad against the exact line -- a property of
the reference's arithmetic that the oracle restates sample by sample (oracle/o_acq.c::advance_phase) and the kernels (closed form in double,
k_symbol8k.hpp::s8_fill_ptab) do not.  CPU test: replays the accumulator and bounds the wander that tests/test_gpu_channel.py::TOL_DEROT allows for."""
import numpy as np


def wander(eps, N, cp, symbols=24):
    inc = -eps / N
    ph, exact, worst = np.float32(0.0), 0.0, 0.0
    two_pi, pi = np.float32(2.0 * np.pi), np.float32(np.pi)
    for _ in range(symbols):
        dev = np.empty(N + cp)
        for i in range(N + cp):
            ph = np.float32(np.float64(ph) + inc)
            exact += inc
            while ph > pi:
                ph -= two_pi
            while ph < -pi:
                ph += two_pi
            dev[i] = (float(ph) - exact + np.pi) % (2 * np.pi) - np.pi
        worst = max(worst, dev.max() - dev.min())          # a constant offset is a common rotation of the symbol: the equaliser divides it out
    return worst


def test_float_phase_accumulator_wanders_within_a_symbol():
    w8 = wander(2.3247788, 8192, 256)                      # epsilon of a +0.37 subcarrier offset
    w2 = wander(2.3247788, 2048, 64)
    assert 1e-4 < w8 < 1.5e-3 and w2 < w8                   # what TOL_DEROT covers; the 2k symbol is a quarter as long
    assert wander(0.0, 8192, 256, symbols=2) == 0.0         # no offset: the accumulator stays at zero (clean loopback taps agree to 1e-6)
