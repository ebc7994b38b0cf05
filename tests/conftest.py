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


def host_example(name):
    """path of a C++ host example under gr_dvbt_amd/host, (re)built when it is missing or older than the header / sources it was compiled from (a stale
    binary has the previous layout of the ABI's structs)"""
    import subprocess
    host = os.path.join(ROOT, "gr_dvbt_amd", "host")
    exe = os.path.join(host, name)
    deps = [os.path.join(ROOT, "include", "dvbt_hip.h"), os.path.join(host, "dvbt_blocks.hpp"), os.path.join(host, name + ".cpp")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.check_call(["bash", os.path.join(host, "build.sh")], stdout=subprocess.DEVNULL)
    return exe
