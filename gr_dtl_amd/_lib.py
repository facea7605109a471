"""ctypes loader for libdtl_hip.so (the C ABI declared in include/dtl_rx.h).

The product path never falls back to a CPU implementation: if the shared library is missing
it raises, and every compute call fails with DTL_ERR_NO_DEVICE when no HIP device is present.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DTL_HIP_LIB") or os.path.join(_HERE, "libdtl_hip.so")  # override: a host-sanitizer build

OK = 0
ERR_INVALID_ARG, ERR_IO, ERR_HIP, ERR_TAGS_MISSING, ERR_CAPACITY, ERR_NO_DEVICE, ERR_UNSUPPORTED, ERR_INTERNAL = (
    -1, -2, -3, -4, -5, -6, -7, -8)


class DtlError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("dtl error %d: %s" % (code, msg))
        self.code = code


def build(force=False):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    script = os.path.join(_HERE, "csrc", "build.sh")
    if force and os.path.isdir(os.path.join(_HERE, "csrc", "_obj")):
        for f in os.listdir(os.path.join(_HERE, "csrc", "_obj")):
            os.remove(os.path.join(_HERE, "csrc", "_obj", f))
    subprocess.run(["bash", script], check=True)
    return LIB_PATH


_lib = None
_vp = C.c_void_p


class CwDesc(C.Structure):
    _fields_ = [("llr_off0", C.c_uint32), ("llr_off1", C.c_uint32), ("out_off", C.c_uint32),
                ("len0", C.c_uint16), ("len1", C.c_uint16), ("k_used", C.c_uint16), ("pad", C.c_uint16)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libdtl_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or gr_dtl_amd/csrc/build.sh -- there is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; importing torch first makes
    # libdtl_hip.so bind to that copy instead of loading /opt/rocm's next to it (two runtimes do not
    # share the device).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    L.dtl_last_error.restype = C.c_char_p
    L.dtl_version.restype = C.c_char_p
    L.dtl_device_count.restype = C.c_int
    L.dtl_ldpc_create.argtypes = [C.c_char_p, C.c_int, C.c_float, C.POINTER(_vp)]
    L.dtl_ldpc_create_from_text.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_float, C.POINTER(_vp)]
    L.dtl_ldpc_destroy.argtypes = [_vp]
    L.dtl_ldpc_destroy.restype = None
    L.dtl_ldpc_set_arith.argtypes = [_vp, C.c_int32]
    L.dtl_ldpc_get_n.argtypes = [_vp]
    L.dtl_ldpc_get_k.argtypes = [_vp]
    L.dtl_ldpc_get_permute.argtypes = [_vp, _vp]
    L.dtl_ldpc_encode_host.argtypes = [_vp, _vp, _vp]
    L.dtl_ldpc_encode_batch_host.argtypes = [_vp, _vp, C.c_size_t, _vp]
    L.dtl_ldpc_decode_batch.argtypes = [_vp, _vp, C.c_size_t, _vp, _vp, _vp]
    L.dtl_ldpc_decode_gather.argtypes = [_vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp]
    _lib = L
    return L


def check(rc):
    if rc != OK:
        raise DtlError(rc, (lib().dtl_last_error() or b"").decode("utf-8", "replace"))
    return rc


def current_stream_ptr():
    """hipStream_t of torch's current stream (0 = default stream)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
