"""shim/ (the GNU Radio side of the drop-in: one gr::block over the C ABI + a fec_dec over dtl_ldpc_*) needs GNU Radio 3.10,
which is not installed here.  Its sources are compiled -- and, on the GPU box, EXECUTED (tests/test_gr_block_gpu.py) --
against tests/cpp/mock_gnuradio/: a functional stand-in for the ~30 GNU Radio names they use (gr::block bookkeeping,
io_signature, pmt, tags, message ports) with GNU Radio 3.10's signatures.  Here, without a GPU: the stand-in passes its own
self-test (a block that breaks a scheduler rule is caught), the block's source compiles and LINKS against libdtl_hip.so
with -Werror, and a wrong C-ABI call still fails to compile.
(The block stands in for the 7-port hier block of python/dtl/ofdm_adaptive_rx.py:42-59.)"""
import os
import subprocess

import pytest

from conftest import ROOT

MOCK = os.path.join(ROOT, "tests", "cpp", "mock_gnuradio")
HIPCC = "/opt/rocm/bin/hipcc"
BUILD = os.path.join(ROOT, "tests", "cpp", "_build")


def _syntax_only(source_text_or_path, is_path=True):
    cmd = [HIPCC, "-fsyntax-only", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-x", "c++",
           "-D__HIP_PLATFORM_AMD__", "-I", MOCK, "-I", os.path.join(ROOT, "shim", "include"), "-I", os.path.join(ROOT, "shim", "lib"),
           "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include"]
    if is_path:
        return subprocess.run(cmd + [source_text_or_path], capture_output=True, text=True, timeout=300)
    return subprocess.run(cmd + ["-"], input=source_text_or_path, capture_output=True, text=True, timeout=300)


def test_mock_gnuradio_passes_its_self_test():
    """gr::mock::harness accounts items / tags / messages like a scheduler and catches the five rule violations"""
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, "test_mock_harness")
    done = subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", MOCK,
                           os.path.join(ROOT, "tests", "cpp", "test_mock_harness.cpp"), "-o", out], capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stderr[-3000:]
    run = subprocess.run([out], capture_output=True, text=True, timeout=60)
    assert run.returncode == 0 and run.stdout.startswith("OK"), run.stdout + run.stderr


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_gr_block_builds_and_links_against_the_library(dtl):
    """the program the GPU test runs: block source + harness driver, -Wall -Wextra -Werror, linked with libdtl_hip.so"""
    from test_gr_block_gpu import build_block_harness
    binary = build_block_harness()
    assert os.path.exists(binary)
    # without a GPU the block's constructor fails loudly (no CPU fallback): the driver reports the exception
    import torch
    if not torch.cuda.is_available():
        run = subprocess.run([binary, "/nonexistent", "/nonexistent", "/tmp", "0", "1", "100"], capture_output=True, text=True, timeout=60)
        assert run.returncode != 0


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
