"""A small randomised parity sweep (tools/sweep.py): random DVB-T parameters (all constellations, code rates, modes, guard
intervals), segment lengths, silence lead-ins (some longer than a window: start-up restarts), Viterbi chunk sizes and noise
levels.  Clean cases must equal the oracle at every integer tap, noisy ones after the RS decoder."""
import os
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_random_sweep():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import sweep
    rng = np.random.RandomState(4321)
    res = [sweep.one(rng, i) for i in range(12)]
    assert all(res)
