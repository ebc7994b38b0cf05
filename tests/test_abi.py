"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every
symbol include/dvbt_hip.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def g():
    import gr_dvbt_amd
    gr_dvbt_amd.build()
    return gr_dvbt_amd


def _declared():
    txt = open(os.path.join(ROOT, "include", "dvbt_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dvbt_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(g):
    L = g.lib()
    names = _declared()
    assert len(names) >= 50
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_one_triple_per_reference_block(g):
    names = set(_declared())
    for blk in ("ofdm_sym_acquisition", "fft", "demod_reference_signals", "demap", "symbol_inner_interleaver",
                "bit_inner_deinterleaver", "viterbi_decoder", "convolutional_deinterleaver", "reed_solomon_dec",
                "energy_descramble"):
        for fn in ("create", "forecast", "work", "destroy"):
            assert f"dvbt_{blk}_{fn}" in names


def test_dims_match_reference_table(g):
    # SURVEY Appendix A
    d = g.get_dims(g.QAM64, g.C7_8, g.T8k)
    assert (d.fft_length, d.cp_length, d.Kmax, d.payload_length, d.zeros_on_left, d.m, d.cr_k, d.cr_n, d.ntraceback) == \
        (8192, 256, 6816, 6048, 688, 6, 7, 8, 24)
    assert d.info_bits_per_symbol == 3969 * 8
    d = g.get_dims(g.QAM16, g.C1_2, g.T2k)
    assert (d.fft_length, d.cp_length, d.payload_length, d.zeros_on_left, d.m, d.ntraceback) == (2048, 64, 1512, 172, 4, 5)
    assert d.info_bits_per_symbol == 378 * 8
    with pytest.raises(g.DvbtError):
        g.get_dims(7, 0, 0)


def test_no_cpu_fallback(g):
    if g.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(g.DvbtError) as e:
        g.Rx(g.QAM16, g.C1_2, g.T2k, max_samples=1 << 20)
    assert "no CPU fallback" in str(e.value)
    L = g.lib()
    h = C.c_void_p()

    class P(C.Structure):
        _fields_ = [("fft_size", C.c_int), ("forward", C.c_int), ("shift", C.c_int)]
    assert L.dvbt_fft_create(C.byref(P(2048, 1, 1)), C.byref(h)) == -2


def test_product_does_not_import_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gr_dvbt_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inc", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in txt and "dvbt_oracle.h" not in txt and "liboracle" not in txt, os.path.join(dirpath, f)


def test_ctypes_structures_have_the_headers_layout(tmp_path):
    """the structures of gr_dvbt_amd/binding.py against include/dvbt_hip.h as a C compiler lays them out (sizes, the offsets of the newest fields): a field added on one
    side only would shift everything behind it silently"""
    import ctypes as C
    import subprocess
    import gr_dvbt_amd.binding as b
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dvbt_hip.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(dvbt_rx_params), offsetof(dvbt_rx_params, viterbi_warm_windows), '
                   'offsetof(dvbt_rx_params, max_samples), sizeof(dvbt_rx_stream_params), offsetof(dvbt_rx_stream_params, segment_superframes), '
                   'offsetof(dvbt_rx_stream_params, borrow_device_pushes), sizeof(dvbt_rx_report)); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(b.RxParams), b.RxParams.viterbi_warm_windows.offset, b.RxParams.max_samples.offset, C.sizeof(b.StreamParams),
            b.StreamParams.segment_superframes.offset, b.StreamParams.borrow_device_pushes.offset, C.sizeof(b.RxReport)]
    assert got == want
