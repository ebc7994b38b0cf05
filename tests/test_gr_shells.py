"""The GNU Radio block shells of gr_dvbt_amd/host/gr: one per receive block, deriving from the reference's
own public class.  GNU Radio is absent here, so this is a syntax check only: g++ -fsyntax-only against the REFERENCE's
include/dvbt/*.h (read in place, never copied) plus declarations of the GNU Radio names used (tests/gr_syntax/).  It proves that
every shell matches its interface's make() signature and overrides forecast/general_work with the right types; it builds and
runs nothing."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GR = os.path.join(ROOT, "gr_dvbt_amd", "host", "gr")
REF_INC = "/root/reference/include"

BLOCKS = ["ofdm_sym_acquisition", "fft_hip", "demod_reference_signals", "dvbt_demap", "symbol_inner_interleaver",
          "bit_inner_deinterleaver", "viterbi_decoder", "convolutional_deinterleaver", "reed_solomon_dec", "energy_descramble", "rx_hip"]


def test_every_receive_block_has_a_shell():
    for b in BLOCKS:
        assert os.path.exists(os.path.join(GR, b + "_impl.h")) and os.path.exists(os.path.join(GR, b + "_impl.cc")), b
    cm = open(os.path.join(GR, "CMakeLists.txt")).read()
    assert "find_package(Gnuradio" in cm and all(b + "_impl.cc" in cm for b in BLOCKS)


@pytest.mark.skipif(not os.path.isdir(REF_INC) or shutil.which("g++") is None, reason="needs the reference's public headers and g++")
@pytest.mark.parametrize("block", BLOCKS)
def test_shell_matches_the_reference_interface(block):
    cmd = ["g++", "-std=gnu++11", "-fsyntax-only", "-Wall", "-Wno-unused", "-I", os.path.join(ROOT, "tests", "gr_syntax"), "-I", REF_INC,
           "-I", os.path.join(GR, "include"), "-I", os.path.join(ROOT, "include"), "-I", GR, os.path.join(GR, block + "_impl.cc")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-3000:]


def test_new_blocks_have_grc_and_swig_glue():
    """dvbt.fft_hip and dvbt.rx_hip are new class names: GRC descriptors and the SWIG lines (pattern: gr-dvbt grc/dvbt_viterbi_decoder.xml:7,
    swig/dvbt_swig.i:53-71) so that they can be placed in a .grc"""
    import xml.etree.ElementTree as ET
    for name, args in (("fft_hip", 3), ("rx_hip", 9)):
        t = ET.parse(os.path.join(GR, "grc", f"dvbt_{name}.xml")).getroot()
        assert t.find("key").text == f"dvbt_{name}" and t.find("import").text == "import dvbt"
        mk = t.find("make").text
        assert mk.startswith(f"dvbt.{name}(") and mk.count("$") == args
        keys = {p.find("key").text for p in t.findall("param")}
        assert all(("$" + k) in mk for k in keys)
    sw = open(os.path.join(GR, "swig", "dvbt_hip_swig.i")).read()
    assert "GR_SWIG_BLOCK_MAGIC2(dvbt, fft_hip);" in sw and "GR_SWIG_BLOCK_MAGIC2(dvbt, rx_hip);" in sw
