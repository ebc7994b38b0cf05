"""bench.py against an A/B or attribution build of the library (same C ABI, other file name): `python tools/ab_bench.py <lib.so> [--wrong-output] <bench args>`.
--wrong-output: the build's results are wrong on purpose (compile-time S8_EXP / V3_EXP attribution hooks): no pre-scan, no verification.
Measurement aids live here; bench.py and the package read no environment variable."""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
so = os.path.abspath(sys.argv[1])
args = sys.argv[2:]
wrong = "--wrong-output" in args
args = [a for a in args if a != "--wrong-output"]
import torch  # noqa: F401,E402
import gr_dvbt_amd.binding as b  # noqa: E402
b._SO = so
import bench  # noqa: E402
bench.EXPERIMENT_BUILD = wrong
sys.argv = ["bench.py"] + args
bench.main()
