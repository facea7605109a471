"""Batched adaptive-OFDM receiver with the reference's hier-block interface.

``ofdm_receiver(config)`` mirrors python/dtl/ofdm_receiver.py:11-56 (same constructor argument, same
derived attributes) and ``ofdm_adaptive_rx.from_parameters(config_dict=None, **kwargs)`` mirrors
python/dtl/ofdm_adaptive_rx.py:32-59.  Instead of wiring GNU Radio blocks, ``work(iq)`` hands a whole
batch of samples (a device tensor) to the HIP pipeline behind the C ABI (include/dtl_rx.h) and returns
the streams the reference hier block exposes on its output ports.
"""
import ctypes as C
import os
import weakref

import numpy as np

from . import _lib
from . import ofdm_adaptive
from .fec import _resolve
from .ofdm_adaptive_config import (constellation_type_t, make_rx_config, ofdm_adaptive_rx_config)


class _Cfg(C.Structure):
    _fields_ = [("fft_len", C.c_int32), ("cp_len", C.c_int32), ("frame_length", C.c_int32), ("has_fec", C.c_int32),
                ("n_occ_sets", C.c_int32), ("max_occ", C.c_int32), ("occ_len", C.c_void_p), ("occ", C.c_void_p),
                ("n_pilot_sets", C.c_int32), ("n_pilots", C.c_int32), ("pilot_carriers", C.c_void_p),
                ("n_pilot_sym_sets", C.c_int32), ("pilot_symbols", C.c_void_p), ("sync_word1", C.c_void_p),
                ("sync_word2", C.c_void_p), ("sync_threshold", C.c_float), ("eq_alpha", C.c_float),
                ("use_sync_correct", C.c_int32), ("scramble_header", C.c_int32), ("llr_dist_mode", C.c_int32),
                ("max_iterations", C.c_int32), ("bp_sigma", C.c_float), ("n_fec_codes", C.c_int32),
                ("alist_texts", C.POINTER(C.c_char_p)), ("alist_lens", C.POINTER(C.c_size_t)),
                ("n_mcs", C.c_int32), ("mcs_snr", C.c_void_p), ("mcs_cnst", C.c_void_p), ("mcs_fec", C.c_void_p),
                ("initial_mcs", C.c_int32), ("decision_th", C.c_int32), ("hysteresis", C.c_double),
                ("arith", C.c_int32), ("device", C.c_int32)]


# enum dtl_arith (include/dtl_rx.h): the reference's operations / the production arithmetic
ARITH = {"gr_exact": 0, "fast": 1, 0: 0, 1: 1}


def make_cfg_struct(cfg):
    """ofdm_adaptive_{rx,tx}_config -> the flattened dtl_rx_cfg of the C ABI (+ the arrays it points to)."""
    fec = len(cfg.fec_codes)
    codes_alist, codes_id = [], {}
    if fec:
        codes_alist = list(zip(*cfg.fec_codes))[1]
        codes_id = {name: id + 1 for (id, name) in enumerate(list(zip(*cfg.fec_codes))[0])}
    mcs = [(snr_th, (cnst, codes_id.get(code_name, 0))) for (snr_th, (cnst, code_name)) in cfg.mcs]
    occ_len = np.array([len(o) for o in cfg.occupied_carriers], dtype=np.int32)
    max_occ = int(occ_len.max())
    occ = np.zeros((len(cfg.occupied_carriers), max_occ), dtype=np.int32)
    for i, o in enumerate(cfg.occupied_carriers):
        occ[i, :len(o)] = o
    # pilot sets may differ in size: padded with DTL_NO_CARRIER / 0
    npil = max([len(p) for p in cfg.pilot_carriers] + [1])
    pil = np.full((len(cfg.pilot_carriers), npil), 0x7fffffff, dtype=np.int32)
    for i, pset in enumerate(cfg.pilot_carriers):
        pil[i, :len(pset)] = list(pset)
    psym = np.zeros((len(cfg.pilot_symbols), npil), dtype=np.complex64)
    for i, row in enumerate(cfg.pilot_symbols):
        psym[i, :len(row)] = list(row)
    sw1 = np.ascontiguousarray(np.asarray(cfg.sync_word1, dtype=np.complex64))
    sw2 = np.ascontiguousarray(np.asarray(cfg.sync_word2, dtype=np.complex64))
    texts = []
    for path in codes_alist:
        with open(_resolve(path), "rb") as f:
            texts.append(f.read())
    n_codes = len(texts)
    arr_texts = (C.c_char_p * max(n_codes, 1))(*texts) if n_codes else (C.c_char_p * 1)()
    arr_lens = (C.c_size_t * max(n_codes, 1))(*[len(t) for t in texts]) if n_codes else (C.c_size_t * 1)()
    mcs_snr = np.array([float(m[0]) for m in mcs], dtype=np.float64)
    mcs_cnst = np.array([int(m[1][0]) for m in mcs], dtype=np.int32)
    mcs_fec = np.array([int(m[1][1]) for m in mcs], dtype=np.int32)
    c = _Cfg()
    c.fft_len, c.cp_len, c.frame_length, c.has_fec = cfg.fft_len, cfg.cp_len, cfg.frame_length, int(bool(fec))
    c.n_occ_sets, c.max_occ = occ.shape
    c.occ_len, c.occ = occ_len.ctypes.data, occ.ctypes.data
    c.n_pilot_sets, c.n_pilots = pil.shape
    c.pilot_carriers = pil.ctypes.data
    c.n_pilot_sym_sets = psym.shape[0]
    c.pilot_symbols = psym.ctypes.data
    c.sync_word1, c.sync_word2 = sw1.ctypes.data, sw2.ctypes.data
    c.sync_threshold = float(getattr(cfg, "sync_threshold", 0.95))
    c.eq_alpha = 0.1                      # ofdm_receiver.py:163
    c.use_sync_correct = int(bool(getattr(cfg, "use_sync_correct", True)))
    c.scramble_header = int(bool(cfg.scramble_bits))
    c.llr_dist_mode = int(getattr(cfg, "llr_dist_mode", 0))
    c.max_iterations = int(getattr(cfg, "ldpc_max_iterations", 15))
    c.bp_sigma = float(getattr(cfg, "ldpc_sigma", 0.0))
    c.n_fec_codes = n_codes
    c.alist_texts = arr_texts
    c.alist_lens = arr_lens
    c.n_mcs = len(mcs)
    c.mcs_snr, c.mcs_cnst, c.mcs_fec = mcs_snr.ctypes.data, mcs_cnst.ctypes.data, mcs_fec.ctypes.data
    c.initial_mcs = int(getattr(cfg, "initial_mcs_id", 0))
    c.decision_th = 5                     # ofdm_receiver.py:167
    c.hysteresis = 1.0
    c.arith = ARITH[getattr(cfg, "arith", "gr_exact")]
    c.device = int(getattr(cfg, "device", -1))   # DTL_DEVICE_CURRENT: the device torch has made current (one process per GPU)
    keep = (occ_len, occ, pil, psym, sw1, sw2, arr_texts, arr_lens, mcs_snr, mcs_cnst, mcs_fec, texts)
    return c, keep


class HeaderFields(C.Structure):   # dtl_header_fields
    _fields_ = [(n, C.c_int32) for n in ("crc_ok", "carr_offset", "payload_len", "frame_no", "cnst", "fb_cnst", "tb_no",
                                         "fec_fb", "tb_offset", "fec", "tb_payload", "pad")]


class FrameEqIn(C.Structure):      # dtl_frame_eq_in
    _fields_ = [("cnst", C.c_int32), ("carr_offset", C.c_int32), ("n_payload_syms", C.c_int32), ("pad", C.c_int32)]


class FrameEqOut(C.Structure):     # dtl_frame_eq_out
    _fields_ = [("snr_db", C.c_double), ("noise_db", C.c_double), ("noise", C.c_double)]


class TbSpan(C.Structure):         # dtl_tb_span
    _fields_ = [("tb_no", C.c_int32), ("fec", C.c_int32), ("tb_payload_bits", C.c_int32), ("n_cw", C.c_int32),
                ("cw_first", C.c_uint32), ("bits_off", C.c_uint32)]


class FrameInfo(C.Structure):
    _fields_ = [("trigger", C.c_int64), ("header_ok", C.c_int32), ("carr_offset", C.c_int32),
                ("payload_len", C.c_int32), ("frame_no", C.c_int32), ("cnst", C.c_int32), ("fb_cnst", C.c_int32),
                ("tb_no", C.c_int32), ("fec_fb", C.c_int32), ("tb_offset", C.c_int32), ("fec", C.c_int32),
                ("tb_payload", C.c_int32), ("n_payload_syms", C.c_int32), ("payload_index", C.c_int32),
                ("snr_db", C.c_double), ("noise", C.c_double), ("feedback_cnst", C.c_int32),
                ("feedback_fec", C.c_int32), ("lost_frames", C.c_int32), ("cand", C.c_int32),
                ("llr_offset", C.c_int64)]


class TbInfo(C.Structure):
    _fields_ = [("tb_no", C.c_int32), ("tb_payload", C.c_int32), ("crc_ok", C.c_int32), ("avg_it", C.c_int32),
                ("user_bits", C.c_int32), ("pad", C.c_int32), ("out_offset", C.c_int64)]


class _Result(C.Structure):
    _fields_ = [("n_frames", C.c_int64), ("n_payload_frames", C.c_int64), ("n_tbs", C.c_int64),
                ("n_payload_bytes", C.c_int64), ("n_raw_triggers", C.c_int64), ("n_fixed_triggers", C.c_int64),
                ("n_llr", C.c_int64), ("n_codewords", C.c_int64),
                ("frames", C.POINTER(FrameInfo)), ("tbs", C.POINTER(TbInfo)),
                ("raw_triggers", C.POINTER(C.c_int64)), ("fixed_triggers", C.POINTER(C.c_int64)),
                ("trigger_angles", C.POINTER(C.c_float)),
                ("d_payload", C.c_void_p), ("d_eq_hard", C.c_void_p), ("d_eq_soft", C.c_void_p),
                ("d_soft_syms", C.c_void_p), ("d_llr", C.c_void_p), ("d_taps", C.c_void_p),
                ("d_hdr_bits", C.c_void_p), ("d_metric", C.c_void_p), ("d_cw_bits", C.c_void_p),
                ("d_cw_nit", C.c_void_p), ("stage_ms", C.c_double * 8),
                ("frame_capacity", C.c_int32), ("hdr_bits", C.c_int32), ("frame_samples", C.c_int32),
                ("pad", C.c_int32)]


WANT = {"eq_hard": 1, "eq_soft": 2, "soft_syms": 4, "hdr_bits": 8, "metric": 16, "kernel_ms": 32, "kernel_split": 64}
KERNEL_NAMES = ("sync_edges_kernel", "header_kernel", "frame_eq_kernel", "demap_kernel", "ldpc_bp_np_kernel", "tb_check_kernel + compaction")


class KernelTimes(C.Structure):   # dtl_rx_kernel_times
    _fields_ = [("ms", C.c_double * 6), ("launches", C.c_int32 * 6), ("longest_launch_ms", C.c_double * 6)]
_FRAME_DTYPE = np.dtype([(n, {C.c_int64: "<i8", C.c_int32: "<i4", C.c_double: "<f8"}[t]) for n, t in FrameInfo._fields_], align=True)
_TB_DTYPE = np.dtype([(n, {C.c_int64: "<i8", C.c_int32: "<i4"}[t]) for n, t in TbInfo._fields_], align=True)


_RECORD_FIELDS = ("frames", "tbs", "raw_triggers", "fixed_triggers", "sc_freq")


def _records(ptr, count, dtype):
    # one memmove (a structured-dtype .copy() goes field by field and costs milliseconds)
    arr = np.empty(count, dtype=dtype)
    if count:
        C.memmove(arr.ctypes.data, ptr, count * arr.itemsize)
    return arr


class rx_result:
    """Outputs of one ``work`` call.  Port numbering follows ofdm_receiver (ofdm_receiver.py:18-20):
    out0 ``payload`` (bytes), out1/out2 trigger positions (fixed / raw), out3 ``eq_hard``,
    out4 ``soft_syms``, out5 ``sc_freq`` (held angle per raw trigger); message ports ``monitor`` /
    ``feedback`` / ``header`` become the per-frame / per-TB record arrays.

    The record arrays (``frames``, ``tbs``, ``raw_triggers``, ``fixed_triggers``, ``sc_freq``) live in host
    memory owned by the C handle (include/dtl_rx.h: valid until the next call on it).  They are copied into
    numpy arrays the first time one of them is read, or -- if this object is still alive then -- just before
    the receiver starts its next call; a result nobody reads costs no copy (~9 MB per 64k-frame batch)."""

    def __init__(self):
        self.payload = None        # torch.uint8 (device)
        self._pending = None       # name -> (pointer, count, dtype) of records still in the handle's memory
        self.eq_hard = self.eq_soft = self.soft_syms = self.llr = self.taps = self.hdr_bits = None
        self.metric = self.cw_bits = self.cw_nit = None
        self.stage_ms = None
        self.n_payload_frames = 0
        self.n_payload_bytes = 0
        self.n_codewords = 0

    def _materialise(self):
        pending, self._pending = self._pending, None
        if pending:
            for name, (ptr, count, dtype) in pending.items():
                self.__dict__[name] = _records(ptr, count, dtype)

    def __getattr__(self, name):   # only reached for attributes not in __dict__
        if name in _RECORD_FIELDS:
            # frames: numpy structured array, one row per visited trigger; tbs: one row per transport block
            self._materialise()
            if name in self.__dict__:
                return self.__dict__[name]
        raise AttributeError(name)

    def payload_bytes(self):
        return self.payload.cpu().numpy()


class ofdm_receiver:
    """Adaptive OFDM Receiver (python/dtl/ofdm_receiver.py:11-56)."""

    def __init__(self, config, name="rx"):
        self.name = name
        self.fft_len = config.fft_len
        self.cp_len = config.cp_len
        self.frame_length_tag_key = config.frame_length_tag_key
        self.packet_length_tag_key = config.packet_length_tag_key
        self.frame_no_tag_key = config.frame_no_tag_key
        self.occupied_carriers = config.occupied_carriers
        self.pilot_carriers = config.pilot_carriers
        self.pilot_symbols = config.pilot_symbols
        self.scramble_bits = config.scramble_bits
        self.rolloff = config.rolloff
        self.sync_threshold = config.sync_threshold
        self.frame_length = config.frame_length
        mc_schemes = list(zip(*config.mcs))[1]
        cnsts = list(zip(*mc_schemes))[0]
        self.constellations = list(set(cnsts))
        self.use_sync_correct = config.use_sync_correct
        self.fec = len(config.fec_codes)
        self.codes_alist = []
        self.codes_id = {}
        if self.fec:
            self.codes_alist = list(zip(*config.fec_codes))[1]
            self.codes_id = {name: id + 1 for (id, name) in enumerate(list(zip(*config.fec_codes))[0])}
        self.mcs = [(snr_th, (cnst, self.codes_id.get(code_name, 0))) for (snr_th, (cnst, code_name)) in config.mcs]
        self.initial_mcs_id = config.initial_mcs_id
        if [self.fft_len, self.fft_len] != [len(config.sync_word1), len(config.sync_word2)]:
            raise ValueError("Length of sync sequence(s) must be FFT length.")
        self.sync_words = [config.sync_word1, config.sync_word2]
        self._config = config
        self._h = C.c_void_p()
        self._last_result = None
        self._setup()

    # -- construction -------------------------------------------------------------------------
    def _setup(self):
        L = _lib.lib()
        self._declare(L)
        self.header_len = 2 if self.fec else 1  # ofdm_receiver.py:66-68
        self.frame_capacity = ofdm_adaptive.frame_capacity(self.frame_length, self.occupied_carriers)
        c, self._keep = make_cfg_struct(self._config)
        _lib.check(L.dtl_rx_create(C.byref(c), C.byref(self._h)))

    @staticmethod
    def _declare(L):
        if getattr(L, "_rx_declared", False):
            return
        L.dtl_rx_create.argtypes = [C.POINTER(_Cfg), C.POINTER(C.c_void_p)]
        L.dtl_rx_destroy.argtypes = [C.c_void_p]
        L.dtl_rx_destroy.restype = None
        L.dtl_rx_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint32, C.c_void_p, C.POINTER(_Result)]
        L.dtl_rx_process_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint32, C.c_int32, C.c_void_p,
                                            C.POINTER(_Result)]
        L.dtl_rx_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint32, C.c_void_p]
        L.dtl_rx_wait.argtypes = [C.c_void_p, C.POINTER(_Result)]
        L.dtl_rx_reset.argtypes = [C.c_void_p]
        L.dtl_rx_reset.restype = None
        L.dtl_rx_stream_position.argtypes = [C.c_void_p]
        L.dtl_rx_stream_position.restype = C.c_int64
        L.dtl_rx_kernel_ms.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.dtl_rx_kernel_split.argtypes = [C.c_void_p, C.POINTER(KernelTimes)]
        L.dtl_copy_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.dtl_chanest.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dtl_header_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(HeaderFields),
                                        C.c_void_p, C.c_void_p, C.c_void_p]
        L.dtl_equalize_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(FrameEqIn), C.c_int64, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.POINTER(FrameEqOut), C.c_void_p]
        L.dtl_tb_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(TbSpan), C.c_int64,
                                    C.POINTER(C.c_int64), C.c_void_p]
        L._rx_declared = True

    def __del__(self):
        try:
            self._settle_last_result()
            if self._h:
                _lib.lib().dtl_rx_destroy(self._h)
        except Exception:
            pass

    @property
    def frame_samples(self):
        return (2 + self.header_len + self.frame_length) * (self.fft_len + self.cp_len)

    # -- single blocks of the graph (frequency-domain vectors in; C ABI dtl_chanest / dtl_header_decode /
    #    dtl_equalize_frames): what per-block shims of ofdm_chanest_vcvc, the header path and
    #    ofdm_adaptive_frame_equalizer_vcvc call --------------------------------------------------------------
    @staticmethod
    def _dev(t, dtype, what):
        if not (hasattr(t, "is_cuda") and t.is_cuda):
            raise _lib.DtlError(_lib.ERR_NO_DEVICE, "%s needs a device tensor (no CPU fallback)" % what)
        if t.dtype != dtype:
            raise ValueError("%s: wrong dtype %s" % (what, t.dtype))
        return t.contiguous()

    def chanest(self, sync_syms):
        """digital.ofdm_chanest_vcvc: sync_syms complex64 [n, 2, fft_len] (FFT output, shifted) ->
        (carrier offsets int32 [n], taps complex64 [n, fft_len]) on the device."""
        import torch
        x = self._dev(sync_syms, torch.complex64, "chanest").view(-1, 2, self.fft_len)
        n = x.shape[0]
        off = torch.empty(n, dtype=torch.int32, device=x.device)
        taps = torch.empty((n, self.fft_len), dtype=torch.complex64, device=x.device)
        _lib.check(_lib.lib().dtl_chanest(self._h, x.data_ptr(), n, off.data_ptr(), taps.data_ptr(), _lib.current_stream_ptr()))
        return off, taps

    def header_decode(self, hdr_syms, taps, carr_offset, want_bits=False):
        """Header equaliser + serializer + BPSK slicer + header parser: hdr_syms complex64 [n, header_len, fft_len],
        taps / carrier offsets from chanest() -> (fields: numpy structured array [n], taps handed to the payload
        equaliser [n, fft_len][, header bits uint8 [n, hdr_bits]])."""
        import torch
        x = self._dev(hdr_syms, torch.complex64, "header_decode").view(-1, self.header_len, self.fft_len)
        n = x.shape[0]
        taps = self._dev(taps, torch.complex64, "header_decode")
        off = self._dev(carr_offset, torch.int32, "header_decode")
        fields = (HeaderFields * max(n, 1))()
        taps_out = torch.empty((n, self.fft_len), dtype=torch.complex64, device=x.device)
        # header symbol i uses carrier set i % n_sets (the kernel's row stride: RxTables.hdr_bits); the sets may differ in size
        occ = self.occupied_carriers
        hdr_bits_len = sum(len(occ[i % len(occ)]) for i in range(self.header_len))
        bits = torch.empty((n, hdr_bits_len), dtype=torch.uint8, device=x.device) if want_bits else None
        _lib.check(_lib.lib().dtl_header_decode(self._h, x.data_ptr(), taps.data_ptr(), off.data_ptr(), n, fields,
                                                bits.data_ptr() if want_bits else None, taps_out.data_ptr(),
                                                _lib.current_stream_ptr()))
        rec = np.frombuffer(bytes(fields), dtype=np.dtype([(k, "<i4") for k, _ in HeaderFields._fields_]))[:n].copy()
        return (rec, taps_out, bits) if want_bits else (rec, taps_out)

    def equalize_frames(self, frames, taps, cnst, carr_offset, n_payload_syms):
        """gr::dtl::ofdm_adaptive_frame_equalizer_vcvc + ofdm_serializer_vcc: frames complex64 [n, frame_length,
        fft_len]; per-frame constellation, carrier offset, payload symbol count (host sequences) ->
        (eq_hard, eq_soft [n, L, fft_len], soft_syms [n, frame_capacity], stats numpy array (snr_db, noise_db, noise))."""
        import torch
        x = self._dev(frames, torch.complex64, "equalize_frames").view(-1, self.frame_length, self.fft_len)
        n = x.shape[0]
        taps = self._dev(taps, torch.complex64, "equalize_frames")
        ins = (FrameEqIn * max(n, 1))()
        for i in range(n):
            ins[i] = FrameEqIn(int(cnst[i]), int(carr_offset[i]), int(n_payload_syms[i]), 0)
        outs = (FrameEqOut * max(n, 1))()
        hard = torch.empty_like(x)
        soft = torch.empty_like(x)
        syms = torch.empty((n, self.frame_capacity), dtype=torch.complex64, device=x.device)
        _lib.check(_lib.lib().dtl_equalize_frames(self._h, x.data_ptr(), taps.data_ptr(), ins, n, hard.data_ptr(),
                                                  soft.data_ptr(), syms.data_ptr(), outs, _lib.current_stream_ptr()))
        stats = np.frombuffer(bytes(outs), dtype=np.dtype([("snr_db", "<f8"), ("noise_db", "<f8"), ("noise", "<f8")]))[:n].copy()
        return hard, soft, syms, stats

    def tb_gather(self, frames, code_nk, device, max_bps=4):
        """gr::dtl::ofdm_adaptive_fec_decoder's transport-block reassembly as a gather table: frames = the `frames`
        record array of a result (header fields + llr_offset), code_nk = [(n, k)] per FEC id 1.. ->
        (dtl_cw_desc table: uint8 device tensor, spans: numpy structured array)."""
        import torch
        fr = np.ascontiguousarray(frames)
        n = len(fr)
        cn = np.array([c[0] for c in code_nk], dtype=np.int32)
        ck = np.array([c[1] for c in code_nk], dtype=np.int32)
        n_cw, n_sp = C.c_int64(0), C.c_int64(0)
        cap_cw = max(16, int(sum(int(f["n_payload_syms"]) * 4 for f in fr) // min(cn) + 4 * n + 16))
        cap_sp = 2 * n + 4
        desc = torch.zeros(cap_cw * C.sizeof(_lib.CwDesc), dtype=torch.uint8, device=device)
        spans = (TbSpan * cap_sp)()
        _lib.check(_lib.lib().dtl_tb_gather(cn.ctypes.data, ck.ctypes.data, len(code_nk), self.frame_capacity, int(max_bps),
                                            fr.ctypes.data, n, desc.data_ptr(), cap_cw, C.byref(n_cw), spans, cap_sp,
                                            C.byref(n_sp), _lib.current_stream_ptr()))
        rec = np.frombuffer(bytes(spans), dtype=np.dtype([("tb_no", "<i4"), ("fec", "<i4"), ("tb_payload_bits", "<i4"),
                                                          ("n_cw", "<i4"), ("cw_first", "<u4"), ("bits_off", "<u4")]))
        return desc[:int(n_cw.value) * C.sizeof(_lib.CwDesc)], rec[:int(n_sp.value)].copy()

    # -- batched work() -----------------------------------------------------------------------
    def work(self, iq, want=(), copy_outputs=True):
        """One self-contained batch.  iq: complex64 CUDA tensor (1-D).  want: any of eq_hard, eq_soft,
        soft_syms, hdr_bits, metric, llr, taps, cw.  Returns rx_result."""
        return self._run(iq, want, copy_outputs, None)

    def feed(self, iq, want=(), final=False, copy_outputs=True):
        """Streaming work(): continues the sample stream of the previous feed() calls (the way the GNU Radio
        scheduler hands consecutive buffers to the reference hier block).  Detector, oscillator, demux,
        transport-block and feedback state persist; samples of incomplete frames stay on the device.
        Trigger positions are absolute.  final=True flushes and resets the stream."""
        return self._run(iq, want, copy_outputs, bool(final))

    def reset(self):
        self._settle_last_result()
        _lib.lib().dtl_rx_reset(self._h)

    def _settle_last_result(self):
        """The previous result's records still sit in the handle's buffers, which the next call overwrites: copy
        them out now if somebody still holds that result."""
        last = self._last_result() if self._last_result is not None else None
        self._last_result = None
        if last is not None:
            last._materialise()

    def kernel_ms(self, back=0):
        """(LDPC kernel ms, phase-B kernels ms) of the call `back` calls ago, from HIP events recorded on that call's
        stream (the handle keeps its last 32 calls); None where the call launched none."""
        a, b = C.c_double(-1.0), C.c_double(-1.0)
        _lib.check(_lib.lib().dtl_rx_kernel_ms(self._h, int(back), C.byref(a), C.byref(b)))
        return (a.value if a.value >= 0 else None, b.value if b.value >= 0 else None)

    def kernel_split(self):
        """Per-kernel device times of the last call made with want=("kernel_split",): {kernel: (ms summed over its
        launches, launches, longest single launch ms)} from HIP events around every launch (dtl_rx_kernel_split)."""
        kt = KernelTimes()
        _lib.check(_lib.lib().dtl_rx_kernel_split(self._h, C.byref(kt)))
        return {KERNEL_NAMES[k]: (float(kt.ms[k]), int(kt.launches[k]), float(kt.longest_launch_ms[k])) for k in range(6)}

    @property
    def stream_position(self):
        return int(_lib.lib().dtl_rx_stream_position(self._h))

    def _run(self, iq, want, copy_outputs, final):
        import torch
        iq, flags = self._check_input(iq, want)
        res = _Result()
        st = _lib.current_stream_ptr()
        self._settle_last_result()
        if final is None:
            _lib.check(_lib.lib().dtl_rx_process(self._h, iq.data_ptr(), iq.numel(), flags, st, C.byref(res)))
        else:
            _lib.check(_lib.lib().dtl_rx_process_stream(self._h, iq.data_ptr(), iq.numel(), flags, int(final), st,
                                                        C.byref(res)))
        return self._result(res, want, copy_outputs, iq.device, iq.numel(), st)

    @staticmethod
    def _check_input(iq, want):
        import torch
        if not (hasattr(iq, "is_cuda") and iq.is_cuda):
            raise _lib.DtlError(_lib.ERR_NO_DEVICE, "ofdm_receiver.work needs a device tensor (no CPU fallback)")
        if iq.dtype != torch.complex64:
            raise ValueError("iq must be complex64")
        flags = 0
        for w in want:
            flags |= WANT.get(w, 0)
        return iq.contiguous().view(-1), flags

    def submit(self, iq, want=()):
        """work() in two halves (dtl_rx_submit / dtl_rx_wait): hand over a self-contained batch -- at most two may wait.  The
        two run side by side (two lanes inside the handle: a receiver, a host thread and a stream each), one's device idle
        is the other's working time.  `iq` is kept alive until its wait()."""
        iq, flags = self._check_input(iq, want)
        if not hasattr(self, "_submitted"):
            self._submitted = []
        # the records of the last result are read from the library's own arrays on first use: a batch handed over now may
        # run on the lane that produced them and overwrite them -- take them first
        self._settle_last_result()
        _lib.check(_lib.lib().dtl_rx_submit(self._h, iq.data_ptr(), iq.numel(), flags, _lib.current_stream_ptr()))
        self._submitted.append((iq, tuple(want)))

    def wait(self, copy_outputs=True):
        """-> the rx_result of the oldest submitted batch (what work() returns for it)"""
        res = _Result()
        self._settle_last_result()
        _lib.check(_lib.lib().dtl_rx_wait(self._h, C.byref(res)))
        iq, want = getattr(self, "_submitted").pop(0)
        return self._result(res, want, copy_outputs, iq.device, iq.numel(), _lib.current_stream_ptr())

    def _result(self, res, want, copy_outputs, dev, n_in, st):
        import torch
        out = rx_result()
        out.n_payload_frames = int(res.n_payload_frames)
        out.n_codewords = int(res.n_codewords)
        out.stage_ms = list(res.stage_ms)
        nf, nt = int(res.n_frames), int(res.n_tbs)
        nr, nx = int(res.n_raw_triggers), int(res.n_fixed_triggers)
        out._pending = {"frames": (res.frames, nf, _FRAME_DTYPE), "tbs": (res.tbs, nt, _TB_DTYPE),
                        "raw_triggers": (res.raw_triggers, nr, np.int64), "fixed_triggers": (res.fixed_triggers, nx, np.int64),
                        "sc_freq": (res.trigger_angles, nr, np.float32)}
        self._last_result = weakref.ref(out)
        out.n_payload_bytes = int(res.n_payload_bytes)
        if not copy_outputs:
            return out

        def grab(ptr, shape, dtype):
            t = torch.empty(shape, dtype=dtype, device=dev)
            if t.numel() and ptr:
                _lib.check(_lib.lib().dtl_copy_device(t.data_ptr(), ptr, t.numel() * t.element_size(), st))
            return t
        out.payload = grab(res.d_payload, (int(res.n_payload_bytes),), torch.uint8)
        npf, Lf, N, cap = out.n_payload_frames, self.frame_length, self.fft_len, int(res.frame_capacity)
        if "eq_hard" in want:
            out.eq_hard = grab(res.d_eq_hard, (npf, Lf, N), torch.complex64)
        if "eq_soft" in want:
            out.eq_soft = grab(res.d_eq_soft, (npf, Lf, N), torch.complex64)
        if "soft_syms" in want:
            out.soft_syms = grab(res.d_soft_syms, (npf, cap), torch.complex64)
        if "llr" in want:
            out.llr = grab(res.d_llr, (int(res.n_llr),), torch.float32)
        if "taps" in want:
            ncand = int(out.frames["cand"].max()) + 1 if nf else 0
            out.taps = grab(res.d_taps, (ncand, N), torch.complex64) if ncand else None
        if "hdr_bits" in want:
            ncand = int(out.frames["cand"].max()) + 1 if nf else 0
            out.hdr_bits = grab(res.d_hdr_bits, (ncand, int(res.hdr_bits)), torch.uint8)
        if "metric" in want:
            out.metric = grab(res.d_metric, (n_in,), torch.float32)
        if "cw" in want and out.n_codewords:
            out.cw_nit = grab(res.d_cw_nit, (out.n_codewords,), torch.int32)
        return out


class rx_ports:
    """The seven output streams of the reference's `ofdm_adaptive_rx` hier block for one buffer of input samples
    (python/dtl/ofdm_adaptive_rx.py:42-59, grc/dtl_ofdm_adaptive_rx.block.yml:22-53), as device tensors, plus the
    `monitor` message port as a list of dicts.  `out[i]` is port i."""

    def __init__(self, out, monitor, result):
        self.out = out
        self.monitor = monitor
        self.result = result          # the underlying rx_result (record arrays)

    def __getitem__(self, i):
        return self.out[i]

    def __len__(self):
        return len(self.out)


class ofdm_adaptive_rx:
    """Adaptive OFDM Simple modem (RX): python/dtl/ofdm_adaptive_rx.py:32-59.  One complex input stream; seven
    output streams + the `monitor` message port:

      0  byte     decoded user bytes (FEC: bytes of the CRC-ok transport blocks; no FEC: every packed frame)
      1  complex  feedback transmitter samples -- the side-channel modem of _setup_feedback_tx (:62-113) is outside
                  the accelerated path: always empty here; the decisions it would send are in monitor / result.frames
      2  byte     trigger stream after ofdm_adaptive_frame_detect_bb: one byte per input sample, 1 at a trigger
      3  byte     trigger stream of ofdm_sync_sc_cfb (raw)
      4  complex x fft_len  payload equaliser out0: equalised and sliced symbols, frame_length vectors per frame
      5  complex  payload equaliser out1 (pre-slice symbols), the vectors flattened to a stream of complex items
                  as the hier block's port declares it (ofdm_receiver.py:18-20,244)
      6  float    fine frequency offset of ofdm_sync_sc_cfb: arg(P) sampled at each raw trigger and held

    work(iq) = one self-contained buffer; feed(iq, final) continues a stream (ofdm_receiver.feed)."""

    @classmethod
    def from_parameters(cls, config_dict=None, **kwargs):
        cfg = make_rx_config(config_dict)
        cfg.__dict__.update(kwargs)          # as the reference does (:38); unknown keys become attributes
        return cls(cfg)

    def __init__(self, config):
        from .monitor import monitor_probe
        self.config = config
        self.direct_rx = ofdm_receiver(config)
        self._probe = monitor_probe()
        self._raw, self._ang, self._fixed = [], [], []   # triggers not yet fully emitted on ports 2 / 3 / 6 (absolute)
        self._emitted = 0                                # ports 2 / 3 / 6 have been produced up to this sample

    def _ports(self, dev, res, lo, hi):
        """Port streams for the sample range [lo, hi) (absolute indices) from the triggers known so far."""
        import torch
        n = max(int(hi - lo), 0)

        def trigger_stream(pos):
            t = torch.zeros(n, dtype=torch.uint8, device=dev)
            p = np.asarray(pos, dtype=np.int64) - lo
            p = p[(p >= 0) & (p < n)]
            if p.size:
                t[torch.as_tensor(p, device=dev)] = 1
            return t
        raw = np.asarray(self._raw, dtype=np.int64)
        ang = np.asarray(self._ang, dtype=np.float32)
        # sample_and_hold_ff: the value at a trigger is held up to the next one; zero before the first
        freq = torch.zeros(n, dtype=torch.float32, device=dev)
        if n and raw.size:
            sel = raw < hi
            pos = torch.as_tensor(raw[sel] - lo, device=dev)
            val = torch.as_tensor(ang[sel], device=dev)
            if pos.numel():
                idx = torch.bucketize(torch.arange(n, device=dev), pos, right=True) - 1
                freq = torch.where(idx >= 0, val[idx.clamp(min=0)], freq)
        N = self.direct_rx.fft_len
        hard = res.eq_hard.reshape(-1, N) if res.eq_hard is not None else torch.zeros((0, N), dtype=torch.complex64, device=dev)
        soft = res.eq_soft.reshape(-1) if res.eq_soft is not None else torch.zeros(0, dtype=torch.complex64, device=dev)
        out = [res.payload, torch.zeros(0, dtype=torch.complex64, device=dev), trigger_stream(self._fixed),
               trigger_stream(self._raw), hard, soft, freq]
        rx = self.direct_rx
        code = None
        if rx.fec:
            from .fec import ldpc_dec
            if not hasattr(self, "_code_nk"):
                d = ldpc_dec(rx.codes_alist[0])
                self._code_nk = (d.get_n(), d.get_k())
            code = self._code_nk
        monitor = self._probe.eq_msgs(res) + (self._probe.dec_msgs(res, *code) if code else self._probe.dec_msgs(res))
        return rx_ports(out, monitor, res)

    def _remember(self, res, keep_from):
        """Triggers of this call joined to the ones still needed: everything at or behind `keep_from`, plus the last
        raw trigger in front of it (its angle is the held value there)."""
        raw = list(self._raw) + [int(v) for v in res.raw_triggers]
        ang = list(self._ang) + [float(v) for v in res.sc_freq]
        first = 0
        for i, p in enumerate(raw):
            if p < keep_from:
                first = i
        self._raw, self._ang = raw[first:], ang[first:]
        self._fixed = [p for p in list(self._fixed) + [int(v) for v in res.fixed_triggers] if p >= keep_from]

    def work(self, iq, want=()):
        """One self-contained buffer of samples -> rx_ports."""
        res = self.direct_rx.work(iq, want=tuple(set(want) | {"eq_hard", "eq_soft"}))
        self._raw, self._ang, self._fixed, self._emitted = [], [], [], 0
        self._remember(res, 0)
        ports = self._ports(iq.device, res, 0, iq.numel())
        self._raw, self._ang, self._fixed = [], [], []
        return ports

    def feed(self, iq, final=False, want=()):
        """Streaming form: consecutive buffers of one sample stream (what the scheduler hands the hier block).  The
        per-sample ports (2, 3, 6) trail the input by one frame, like the reference graph's own latency: a trigger is
        known only when its plateau has ended and frame_detect_bb has seen the following samples.  final=True flushes."""
        end = self.direct_rx.stream_position + iq.numel()
        res = self.direct_rx.feed(iq, want=tuple(set(want) | {"eq_hard", "eq_soft"}), final=final)
        hi = end if final else max(self._emitted, end - self.direct_rx.frame_samples)
        self._remember(res, self._emitted)
        ports = self._ports(iq.device, res, self._emitted, hi)
        self._emitted = hi
        self._remember(_NoTriggers, hi)
        if final:
            self._raw, self._ang, self._fixed, self._emitted = [], [], [], 0
        return ports


class _NoTriggers:
    raw_triggers = fixed_triggers = sc_freq = ()
