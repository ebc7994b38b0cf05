import os
import sys
import pytest

# PyTorch bundles its own HIP runtime; when both it and libdvbt_hip.so live in one process the runtime
# that is loaded first must be torch's (bench.py imports torch first as well), otherwise torch reports
# "No HIP GPUs are available".
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def po():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle
