"""LDPC FEC objects with the reference's interface.

Mirrors include/gnuradio/dtl/fec.h:25-37 (``fec_dec{decode, get_k, get_n}``,
``make_ldpc_decoders`` with index 0 == None meaning "no FEC", lib/dtl/ldpc_dec.cc:23-31) as
bound to Python in python/dtl/bindings/fec_python.cc, on top of the HIP C ABI.  Batched entry
points (``decode_batch`` / ``decode_gather``) operate on device tensors.
"""
import ctypes as C
import os

import numpy as np

from . import _lib

CODES_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "codes")


def _resolve(path):
    if os.path.exists(path):
        return path
    cand = os.path.join(CODES_DIR, os.path.basename(path))
    if os.path.exists(cand):
        return cand
    return path


class ldpc_dec:
    """gr::dtl::ldpc_dec (lib/dtl/ldpc_dec.h:19-38): alist file + max iterations (15)."""

    def __init__(self, alist_fname, max_it=15, sigma=0.0, arith="gr_exact"):
        self._h = C.c_void_p()
        L = _lib.lib()
        _lib.check(L.dtl_ldpc_create(os.fsencode(_resolve(alist_fname)), int(max_it), float(sigma),
                                     C.byref(self._h)))
        self.set_arith(arith)
        self._n = L.dtl_ldpc_get_n(self._h)
        self._k = L.dtl_ldpc_get_k(self._h)

    def __del__(self):
        try:
            if self._h:
                _lib.lib().dtl_ldpc_destroy(self._h)
        except Exception:
            pass

    def set_arith(self, arith):
        """"gr_exact" (0): awgn_bp's operations, the reference's bits; "fast" (1): production arithmetic."""
        _lib.check(_lib.lib().dtl_ldpc_set_arith(self._h, {"gr_exact": 0, "fast": 1}.get(arith, arith)))

    def get_k(self):
        return self._k

    def get_n(self):
        return self._n

    def permute(self):
        out = np.zeros(self._n, dtype=np.int32)
        _lib.check(_lib.lib().dtl_ldpc_get_permute(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def encode(self, data_bits):
        """fec_enc::encode (lib/dtl/ldpc_enc.cc:58-68), host side: k bits -> n bits."""
        data = np.ascontiguousarray(data_bits, dtype=np.uint8)
        if data.size != self._k:
            raise ValueError("encode expects k=%d bits" % self._k)
        out = np.zeros(self._n, dtype=np.uint8)
        _lib.check(_lib.lib().dtl_ldpc_encode_host(self._h, data.ctypes.data_as(C.c_void_p),
                                                  out.ctypes.data_as(C.c_void_p)))
        return out

    def encode_batch(self, data_bits):
        """ncw x k bits -> ncw x n bits (host)."""
        data = np.ascontiguousarray(data_bits, dtype=np.uint8).reshape(-1, self._k)
        out = np.zeros((data.shape[0], self._n), dtype=np.uint8)
        _lib.check(_lib.lib().dtl_ldpc_encode_batch_host(self._h, data.ctypes.data_as(C.c_void_p), data.shape[0],
                                                        out.ctypes.data_as(C.c_void_p)))
        return out

    # -- device paths -------------------------------------------------------------------
    def decode_batch(self, llr, out_bits=None, out_nit=None):
        """llr: CUDA float32 tensor [ncw, n] -> (bits uint8 [ncw, k], nit int32 [ncw])."""
        import torch
        if not llr.is_cuda:
            raise _lib.DtlError(_lib.ERR_NO_DEVICE, "decode_batch needs a device tensor (no CPU fallback)")
        llr = llr.contiguous().view(-1, self._n)
        if llr.dtype != torch.float32:
            raise ValueError("llr must be float32")
        ncw = llr.shape[0]
        if out_bits is None:
            out_bits = torch.empty((ncw, self._k), dtype=torch.uint8, device=llr.device)
        if out_nit is None:
            out_nit = torch.empty((ncw,), dtype=torch.int32, device=llr.device)
        _lib.check(_lib.lib().dtl_ldpc_decode_batch(self._h, llr.data_ptr(), ncw, out_bits.data_ptr(),
                                                   out_nit.data_ptr(), _lib.current_stream_ptr()))
        return out_bits, out_nit

    def decode_gather(self, llr_stream, desc, out_bits, out_nit):
        """Gather-table form (tb_decoder::decode): desc = uint8 CUDA tensor holding dtl_cw_desc[]."""
        ncw = desc.numel() // C.sizeof(_lib.CwDesc)
        _lib.check(_lib.lib().dtl_ldpc_decode_gather(self._h, llr_stream.data_ptr(), desc.data_ptr(), ncw,
                                                    out_bits.data_ptr(), out_nit.data_ptr(),
                                                    _lib.current_stream_ptr()))
        return out_bits, out_nit

    def decode(self, in_data):
        """fec_dec::decode(const float*, int* nit, unsigned char*) for ONE codeword held on the
        host: returns (bits[k], nit).  Goes through the device (H2D, kernel, D2H)."""
        import torch
        llr = torch.as_tensor(np.ascontiguousarray(in_data, dtype=np.float32)).view(1, -1).cuda()
        bits, nit = self.decode_batch(llr)
        return bits.cpu().numpy()[0], int(nit.cpu()[0])


class ldpc_enc(ldpc_dec):
    """gr::dtl::ldpc_enc (lib/dtl/ldpc_enc.h): same code object, host-side encode."""


def make_ldpc_encoders(alist_fnames):
    """lib/dtl/ldpc_enc.cc:23-32"""
    return [None] + [ldpc_enc(f) for f in alist_fnames]


def make_ldpc_decoders(alist_fnames, max_it=15, arith="gr_exact"):
    """lib/dtl/ldpc_dec.cc:23-31: index 0 is None ("no FEC")."""
    return [None] + [ldpc_dec(f, max_it, arith=arith) for f in alist_fnames]
