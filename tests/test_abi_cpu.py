"""The C-ABI shared library builds for gfx950, loads without a GPU, exports every symbol that
include/dtl_rx.h declares, and refuses to compute without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ALIST_300, ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dtl_rx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dtl_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(dtl):
    L = dtl._lib.lib()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libdtl_hip.so does not export %s" % n
    assert b"gfx950" in L.dtl_version()


def test_code_object_is_gfx950(dtl):
    blob = open(dtl._lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for kernel in (b"sync_edges_kernel", b"header_kernel", b"frame_eq_kernel", b"demap_kernel", b"tb_check_kernel",
                   b"ldpc_bp_np_kernel", b"tx_encode_kernel", b"tx_frame_kernel"):
        assert kernel in blob, kernel


def test_no_cpu_fallback_without_device(dtl):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    dec = dtl.ldpc_dec(ALIST_300)
    with pytest.raises(Exception):
        dec.decode(np.zeros(300, np.float32))
    L = dtl._lib.lib()
    h = C.c_void_p()
    assert L.dtl_ldpc_create(os.fsencode(ALIST_300), 15, C.c_float(0), C.byref(h)) == 0
    buf = (C.c_float * 300)()
    rc = L.dtl_ldpc_decode_batch(h, buf, 1, buf, buf, None)
    assert rc == dtl._lib.ERR_NO_DEVICE
    assert b"no CPU fallback" in L.dtl_last_error()
    rx = dtl.ofdm_receiver(dtl.ofdm_adaptive_rx_config())
    with pytest.raises(dtl._lib.DtlError):
        rx.work(torch.zeros(10, dtype=torch.complex64))
    L.dtl_ldpc_destroy(h)


def test_error_codes_and_bad_arguments(dtl):
    L = dtl._lib.lib()
    h = C.c_void_p()
    assert L.dtl_ldpc_create(b"/nonexistent.alist", 15, C.c_float(0), C.byref(h)) == dtl._lib.ERR_IO
    assert L.dtl_ldpc_create_from_text(b"3 2\n", 4, 15, C.c_float(0), C.byref(h)) == dtl._lib.ERR_INVALID_ARG
    cfg = dtl.ofdm_adaptive_rx_config()
    cfg.sync_word1 = cfg.sync_word1[:10]
    with pytest.raises(ValueError, match="Length of sync sequence"):
        dtl.ofdm_receiver(cfg)
