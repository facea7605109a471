"""The kernel-side double exp / log10 / 10^y (dtl_math.h dm_exp64: the LDPC decoder's channel likelihood ratios; dm_log10_64,
dm_pow10_64: the frame's SNR and noise figures) return the bits of the oracle's orc_exp / orc_log10 / orc_pow10 -- exp on
44 million float arguments and around the range edges, the other two on 6 million random doubles each plus every binade
and the special values: compiled with hipcc at test time, the GPU evaluates, the host compares."""
import os
import subprocess

import pytest

from conftest import ROOT


def _build():
    out_dir = os.path.join(ROOT, "tests", "cpp", "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "test_exp64")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off",
           os.path.join(ROOT, "tests", "cpp", "test_exp64.hip"), "-o", out]
    subprocess.run(cmd, check=True, capture_output=True)
    return out


def test_exp64_check_compiles():
    assert os.path.exists(_build())


@pytest.mark.gpu
def test_exp64_gpu_equals_oracle(cuda):
    out = subprocess.run([_build()], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("OK checked="), out.stdout + out.stderr
