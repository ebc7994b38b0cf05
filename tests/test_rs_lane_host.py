"""rs_decode_word_lane (k_backend.hpp), the word-per-lane RS decoder used when most words of a wavefront carry errors, compiled for the host and
compared with the oracle's decoder on 40,000 words: return value and the 188 output bytes, correctable, uncorrectable and garbage words, both modes
(the intended decoder and the as-compiled reference quirk, SURVEY B-1)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lane_decoder_equals_the_oracle(tmp_path, po):
    src = open(os.path.join(ROOT, "gr_dvbt_amd", "csrc", "k_backend.hpp")).read()
    a, b = src.index("__device__ inline int rs_decode_word_lane"), src.index("constexpr int RS_LANE_MIN")
    (tmp_path / "lane_fn.inc").write_text(src[a:b])
    exe = str(tmp_path / "rs_lane_host")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", str(tmp_path), "-o", exe, os.path.join(ROOT, "tests", "rs_host", "rs_lane_host.cpp"),
                           os.path.join(ROOT, "oracle", "liboracle.so"), "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    out = subprocess.check_output([exe], text=True)
    assert out.strip().endswith("0 mismatches of 40000"), out
