"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package (gr_dtl_amd).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")


def build(force=False):
    """Compile the C restatement with the committed Makefile (gcc only)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


c32 = np.complex64
_P = C.c_void_p


def _ptr(a):
    return a.ctypes.data_as(_P)


def _declare(L):
    L.orc_repack_lsb_first.restype = C.c_int
    L.orc_repack_lsb_first.argtypes = [_P, C.c_size_t, _P, C.c_int, C.c_int, C.c_int]
    L.orc_repack_msb_first.restype = C.c_int
    L.orc_repack_msb_first.argtypes = [_P, C.c_size_t, _P, C.c_int, C.c_int, C.c_int]
    L.orc_crc_compute.restype = C.c_uint64
    L.orc_crc_compute.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int,
                                  _P, C.c_size_t]
    L.orc_crc32_append.restype = C.c_uint32
    L.orc_crc32_append.argtypes = [_P, C.c_size_t]
    L.orc_crc32_verify.restype = C.c_int
    L.orc_crc32_verify.argtypes = [_P, C.c_size_t]
    L.orc_bits_per_symbol.restype = C.c_int
    L.orc_bits_per_symbol.argtypes = [C.c_int]
    L.orc_constellation_points.restype = C.c_int
    L.orc_constellation_points.argtypes = [C.c_int, _P]
    L.orc_soft_demap.restype = None
    L.orc_soft_demap.argtypes = [C.c_int, _P, C.c_int, C.c_double, C.c_int, _P]
    L.orc_ldpc_load.restype = _P
    L.orc_ldpc_load.argtypes = [C.c_char_p]
    L.orc_ldpc_free.argtypes = [_P]
    for f in ("orc_ldpc_n", "orc_ldpc_k", "orc_ldpc_m", "orc_ldpc_nedges"):
        getattr(L, f).restype = C.c_int
        getattr(L, f).argtypes = [_P]
    L.orc_ldpc_permute.argtypes = [_P, _P]
    L.orc_ldpc_parity_matrix.argtypes = [_P, _P]
    L.orc_ldpc_encode.argtypes = [_P, _P, _P]
    L.orc_ldpc_decode.restype = C.c_int
    L.orc_ldpc_decode.argtypes = [_P, _P, C.c_int, C.c_float, _P, _P]
    L.orc_ldpc_decode_batch.argtypes = [_P, _P, C.c_int, C.c_int, C.c_float, _P, _P, C.c_int]


# ---------------------------------------------------------------- bits / crc

def repack_lsb_first(data, bits_in, bits_out, full_output_symbols=False):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    out = np.zeros(data.size * bits_in // bits_out + 2, dtype=np.uint8)
    n = lib().orc_repack_lsb_first(_ptr(data), data.size, _ptr(out), bits_in, bits_out,
                                   int(full_output_symbols))
    return out[:n].copy()


def repack_msb_first(data, bits_in, bits_out, full_output_symbols=False):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    out = np.zeros(data.size * bits_in // bits_out + 2, dtype=np.uint8)
    n = lib().orc_repack_msb_first(_ptr(data), data.size, _ptr(out), bits_in, bits_out,
                                   int(full_output_symbols))
    return out[:n].copy()


def crc_compute(num_bits, poly, init, final_xor, refin, refout, data):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    return int(lib().orc_crc_compute(num_bits, poly, init, final_xor, int(refin), int(refout),
                                     _ptr(data), data.size))


def crc32_append(data):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    buf = np.concatenate([data, np.zeros(4, np.uint8)])
    lib().orc_crc32_append(_ptr(buf), data.size)
    return buf


def crc32_verify(buf):
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    return bool(lib().orc_crc32_verify(_ptr(buf), buf.size))


# ---------------------------------------------------------------- constellations

BPSK, QPSK, PSK8, QAM16 = 1, 2, 3, 4


def bits_per_symbol(cnst):
    return lib().orc_bits_per_symbol(int(cnst))


def constellation_points(cnst):
    pts = np.zeros(16, dtype=c32)
    n = lib().orc_constellation_points(int(cnst), _ptr(pts))
    return pts[:n].copy()


def decision_maker(cnst, samples):
    L = lib()
    L.orc_decision_maker.restype = C.c_uint

    class _c(C.Structure):
        _fields_ = [("re", C.c_float), ("im", C.c_float)]
    L.orc_decision_maker.argtypes = [C.c_int, _c]
    s = np.atleast_1d(np.asarray(samples, dtype=c32))
    return np.array([L.orc_decision_maker(int(cnst), _c(float(x.real), float(x.imag))) for x in s],
                    dtype=np.uint8)


def soft_demap(cnst, syms, noise=None, dist_mode=0):
    syms = np.ascontiguousarray(syms, dtype=c32)
    bps = bits_per_symbol(cnst)
    out = np.zeros(syms.size * bps, dtype=np.float32)
    with np.errstate(all="ignore"):
        lib().orc_soft_demap(int(cnst), _ptr(syms), syms.size,
                             -1.0 if noise is None else float(noise), int(dist_mode), _ptr(out))
    return out


# ---------------------------------------------------------------- LDPC

class Ldpc:
    def __init__(self, alist_path):
        self._h = lib().orc_ldpc_load(os.fsencode(alist_path))
        if not self._h:
            raise IOError("cannot load alist %r" % (alist_path,))
        L = lib()
        self.n, self.k, self.m, self.nedges = (L.orc_ldpc_n(self._h), L.orc_ldpc_k(self._h),
                                               L.orc_ldpc_m(self._h), L.orc_ldpc_nedges(self._h))
        self.rank = self.n - self.k
        self.permute = np.zeros(self.n, dtype=np.int32)
        L.orc_ldpc_permute(self._h, _ptr(self.permute))

    def __del__(self):
        try:
            lib().orc_ldpc_free(self._h)
        except Exception:
            pass

    def parity_matrix(self):
        out = np.zeros((self.rank, self.k), dtype=np.uint8)
        lib().orc_ldpc_parity_matrix(self._h, _ptr(out))
        return out

    def encode(self, data_bits):
        data = np.ascontiguousarray(data_bits, dtype=np.uint8)
        assert data.size == self.k
        out = np.zeros(self.n, dtype=np.uint8)
        lib().orc_ldpc_encode(self._h, _ptr(data), _ptr(out))
        return out

    def decode(self, llr, max_it=15, sigma=np.sqrt(2.0)):
        llr = np.ascontiguousarray(llr, dtype=np.float32)
        assert llr.size == self.n
        out = np.zeros(self.k, dtype=np.uint8)
        nit = C.c_int(0)
        lib().orc_ldpc_decode(self._h, _ptr(llr), max_it, float(np.float32(sigma)), _ptr(out),
                              C.byref(nit))
        return out, nit.value

    def decode_batch(self, llr, max_it=15, sigma=np.sqrt(2.0), n_threads=1):
        llr = np.ascontiguousarray(llr, dtype=np.float32).reshape(-1, self.n)
        ncw = llr.shape[0]
        out = np.zeros((ncw, self.k), dtype=np.uint8)
        nit = np.zeros(ncw, dtype=np.int32)
        lib().orc_ldpc_decode_batch(self._h, _ptr(llr), ncw, max_it, float(np.float32(sigma)),
                                    _ptr(out), _ptr(nit), n_threads)
        return out, nit
