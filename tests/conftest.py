import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CODES = os.path.join(ROOT, "gr_dtl_amd", "codes")
ALIST_300 = os.path.join(CODES, "n_0300_k_0152_gap_03.alist")
ALIST_100_23 = os.path.join(CODES, "n_0100_k_0023_gap_10.alist")
ALIST_100_27 = os.path.join(CODES, "n_0100_k_0027_gap_04.alist")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def dtl():
    import gr_dtl_amd
    if not os.path.exists(gr_dtl_amd._lib.LIB_PATH):
        gr_dtl_amd._lib.build()
    return gr_dtl_amd


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return torch.device("cuda:0")
