"""shim/ (the GNU Radio side of the drop-in: one gr::block over the C ABI + a fec_dec over dtl_ldpc_*) cannot be built
here -- GNU Radio 3.10 is not installed -- so its sources had never been through a compiler.  This is a SYNTAX check,
not a GNU Radio test: the files are compiled with -fsyntax-only against tests/cpp/mock_gnuradio/, which declares the
~25 GNU Radio names they use with GNU Radio 3.10's signatures (gr::block, io_signature, pmt, tags, message ports).
It catches typos, wrong argument lists against include/dtl_rx.h and missing includes; it says nothing about behaviour.
(The block stands in for the 7-port hier block of python/dtl/ofdm_adaptive_rx.py:42-59.)"""
import os
import subprocess

import pytest

from conftest import ROOT

MOCK = os.path.join(ROOT, "tests", "cpp", "mock_gnuradio")
HIPCC = "/opt/rocm/bin/hipcc"


def _syntax_only(source_text_or_path, is_path=True):
    cmd = [HIPCC, "-fsyntax-only", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-x", "c++",
           "-D__HIP_PLATFORM_AMD__", "-I", MOCK, "-I", os.path.join(ROOT, "shim", "include"), "-I", os.path.join(ROOT, "shim", "lib"),
           "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include"]
    if is_path:
        return subprocess.run(cmd + [source_text_or_path], capture_output=True, text=True, timeout=300)
    return subprocess.run(cmd + ["-"], input=source_text_or_path, capture_output=True, text=True, timeout=300)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_gr_block_source_passes_a_syntax_check_against_mock_headers():
    out = _syntax_only(os.path.join(ROOT, "shim", "lib", "ofdm_adaptive_rx_hip_impl.cc"))
    assert out.returncode == 0, out.stderr[-4000:]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_fec_dec_shim_header_passes_a_syntax_check_against_mock_headers():
    src = '#include "ldpc_dec_hip.h"\nint main() { auto d = gr::dtl::make_ldpc_decoders_hip({"a.alist"}); return int(d.size()); }\n'
    out = _syntax_only(src, is_path=False)
    assert out.returncode == 0, out.stderr[-4000:]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_the_check_does_catch_a_wrong_call():
    """the mock is not so loose that anything compiles: a C-ABI call with a missing argument must fail"""
    src = ('#include "ofdm_adaptive_rx_hip_impl.h"\n'
           'int f(dtl_rx* h, const float* x) { dtl_rx_result r; return dtl_rx_process_stream(h, x, 10, 0, nullptr, &r); }\n')
    out = _syntax_only(src, is_path=False)
    assert out.returncode != 0 and "dtl_rx_process_stream" in out.stderr
