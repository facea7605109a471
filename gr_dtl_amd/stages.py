"""Single stages of the receive graph on the GPU (C ABI: dtl_sync_sc, dtl_fft_batch, dtl_soft_demap, dtl_crc32): what a
per-block shim of the reference graph (ofdm_sync_sc_cfb, fft_vcc, ofdm_adaptive_constellation_soft_cf, crc_util) would
call.  The receiver itself runs fused kernels with the same device arithmetic."""
import ctypes as C

import numpy as np

from . import _lib


def _declare(L):
    if getattr(L, "_stages_declared", False):
        return
    L.dtl_sync_sc.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]
    L.dtl_fft_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
    L.dtl_soft_demap.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_int32, C.c_void_p, C.c_void_p]
    L.dtl_crc32.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_uint32), C.c_void_p]
    L._stages_declared = True


def _need_device(t, what):
    if not (hasattr(t, "is_cuda") and t.is_cuda):
        raise _lib.DtlError(_lib.ERR_NO_DEVICE, "%s needs a device tensor (no CPU fallback)" % what)


def sync_sc(iq, fft_len, cp_len, threshold=0.95, want_metric=False):
    """-> (trigger positions int64, held angles float32[, metric tensor])"""
    import torch
    _need_device(iq, "sync_sc")
    L = _lib.lib()
    _declare(L)
    iq = iq.contiguous().view(-1)
    n = iq.numel()
    cap = max(16, n // max(fft_len, 1) + 16)
    trig = np.zeros(cap, np.int64)
    ang = np.zeros(cap, np.float32)
    n_out = C.c_int64(0)
    metric = torch.empty(n, dtype=torch.float32, device=iq.device) if want_metric else None
    _lib.check(L.dtl_sync_sc(iq.data_ptr(), n, int(fft_len), int(cp_len), float(threshold),
                             metric.data_ptr() if want_metric else None, trig.ctypes.data, ang.ctypes.data, cap,
                             C.byref(n_out), _lib.current_stream_ptr()))
    k = int(n_out.value)
    return (trig[:k].copy(), ang[:k].copy(), metric) if want_metric else (trig[:k].copy(), ang[:k].copy())


def fft_batch(x, forward=True):
    """x: complex64 device tensor [..., fft_len]; GR fft_vcc with shift=True, unnormalised."""
    import torch
    _need_device(x, "fft_batch")
    L = _lib.lib()
    _declare(L)
    x = x.contiguous()
    out = torch.empty_like(x)
    n_vec = x.numel() // x.shape[-1] if x.numel() else 0
    _lib.check(L.dtl_fft_batch(x.data_ptr(), out.data_ptr(), n_vec, int(x.shape[-1]), int(bool(forward)),
                               _lib.current_stream_ptr()))
    return out


DEMAP_FAST = 0x100   # DTL_DEMAP_FAST


def soft_demap(syms, constellation, noise=None, dist_mode=0, arith="gr_exact"):
    """syms: complex64 device tensor (1-D) -> float32 soft bits, bits/symbol per symbol, LSB first.
    arith "fast": the separable 16-QAM form (DTL_DEMAP_FAST)."""
    import torch
    if arith in ("fast", 1):
        dist_mode = int(dist_mode) | DEMAP_FAST
    _need_device(syms, "soft_demap")
    L = _lib.lib()
    _declare(L)
    syms = syms.contiguous().view(-1)
    bps = int(constellation) if int(constellation) < 4 else 4
    out = torch.empty(syms.numel() * bps, dtype=torch.float32, device=syms.device)
    _lib.check(L.dtl_soft_demap(syms.data_ptr(), syms.numel(), int(constellation), -1.0 if noise is None else float(noise),
                                int(dist_mode), out.data_ptr(), _lib.current_stream_ptr()))
    return out


def crc32(data):
    """data: uint8 device tensor -> CRC-32 value (int)"""
    _need_device(data, "crc32")
    L = _lib.lib()
    _declare(L)
    data = data.contiguous().view(-1)
    v = C.c_uint32(0)
    _lib.check(L.dtl_crc32(data.data_ptr(), data.numel(), C.byref(v), _lib.current_stream_ptr()))
    return int(v.value)
