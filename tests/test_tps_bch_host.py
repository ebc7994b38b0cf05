"""The TPS word's BCH check as the segment-parallel TPS bookkeeping runs it (bch_check_tab on the table the host builds once, k_frontend.hpp) against the
oracle's bit-serial restatement of verify_bch_code: formatted TPS words of every configuration and frame, the same with 1..4 flipped bits, random words."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_table_driven_bch_check_equals_the_oracle(tmp_path, po):
    src = open(os.path.join(ROOT, "gr_dvbt_amd", "csrc", "k_frontend.hpp")).read()
    a = src.index("__device__ __forceinline__ int bch_check_tab")
    b = src.index("// One symbol of the bookkeeping on registers")
    (tmp_path / "bch_fn.inc").write_text(src[a:b])
    exe = str(tmp_path / "tps_bch_host")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", str(tmp_path), "-I", os.path.join(ROOT, "oracle"), "-o", exe,
                           os.path.join(ROOT, "tests", "host", "tps_bch_host.cpp"), os.path.join(ROOT, "oracle", "liboracle.so"),
                           "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    out = subprocess.check_output([exe], text=True)
    m = re.match(r"(\d+) words, (\d+) valid, (\d+) mismatches", out.strip())
    assert m and int(m.group(3)) == 0 and int(m.group(2)) >= 1900, out
