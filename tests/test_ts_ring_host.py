"""dvbt::TsRing (gr_dvbt_amd/csrc/ts_ring.hpp): where the streaming entry puts its decoded chunks inside the pinned output ring.  The ring is 96 MB, a test
stream never fills it, so its turning around is checked here on the CPU: the header compiled with g++, random placements and in-order releases on rings of a
few KB, every chunk's bytes verified when it leaves."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ring_placement_random_sequences(tmp_path):
    exe = str(tmp_path / "ts_ring_host")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "gr_dvbt_amd", "csrc"), "-o", exe, os.path.join(ROOT, "tests", "host", "ts_ring_host.cpp")])
    out = subprocess.check_output([exe], text=True)
    m = re.match(r"(\d+) placements, (\d+) in the ring, (\d+) wrap-arounds, (\d+) errors", out.strip())
    assert m, out
    assert int(m.group(4)) == 0, out
    assert int(m.group(2)) > 50000 and int(m.group(3)) > 3000, out      # the sequences did exercise the ring and its turning around
