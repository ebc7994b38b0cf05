"""gr_dvbt_amd/csrc/k_drift.hpp on the CPU: the closed form of the reference's float phase accumulator (time potential per increment, binade by binade)
and the parallel fixed point the kernels use for the phase at every call entry, against the literal accumulator (one float addition per sample,
ofdm_sym_acquisition_impl.cc:285-309).  tests/drift/drift_host.cpp includes the header's host-callable arithmetic and replays the kernels' scheme."""
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("drift") / "drift_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", out, os.path.join(ROOT, "tests", "drift", "drift_host.cpp")])
    return out


def run(exe, N, cp, nsym, eps, jitter, seed=1):
    o = subprocess.check_output([exe, str(N), str(cp), str(nsym), repr(eps), repr(jitter), str(seed)], text=True).split()
    return tuple(float(x) for x in o)          # literal wander, residual of the sequential closed form, of the parallel scheme, entry-phase error of the latter


@pytest.mark.parametrize("N,cp,nsym,eps,jitter", [
    (8192, 256, 1500, 2.3247788, 1e-6),       # +0.37 subcarriers, clean loopback
    (8192, 256, 1500, -1.2566, 1e-6),         # -0.2
    (8192, 256, 1200, 0.1, 1e-7),
    (8192, 256, 1200, 0.0167, 1e-7),          # a small constant offset: the accumulator crawls through the coarse binades
    (8192, 256, 1200, 0.3, 1e-3),             # estimates that jitter (echoes)
    (8192, 256, 1200, -2.32, 0.02),           # ... a lot (noise)
    (2048, 64, 4000, 2.3247788, 1e-6),
    (2048, 64, 3000, 0.05, 1e-4),
    (8192, 1024, 600, 1.0, 1e-4),             # GI 1/8
])
def test_closed_form_reproduces_the_float_accumulator(exe, N, cp, nsym, eps, jitter):
    lit, seq, par, ent = run(exe, N, cp, nsym, eps, jitter)
    assert lit > 5e-5                          # there is something to reproduce
    assert seq < 1e-5 and par < 1e-5           # two orders of magnitude below the wander, well inside the EQ tap's tolerance
    assert ent < 1e-3
