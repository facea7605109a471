"""Batched adaptive-OFDM transmitter -- the synthetic-frame generator.

Two implementations of the same graph behind one class: `work()` (host, numpy) and `work_device()` (HIP
kernels behind dtl_tx_* of the C ABI: framing state machine on lengths on the host, CRC-32 / LDPC encoding /
mapping / allocation / IFFT / cyclic prefix on the GPU; frames are bit-identical to the CPU oracle's TX).

Mirrors the TX graph of python/dtl/ofdm_transmitter.py:63-189 for one large PDU and a constant MCS:
ofdm_adaptive_fec_frame_bvb (lib/dtl/ofdm_adaptive_fec_frame_bvb_impl.cc:284-565) or
ofdm_adaptive_frame_bb (lib/dtl/ofdm_adaptive_frame_bb_impl.cc:131-163), ofdm_adaptive_packet_header
(lib/dtl/ofdm_adaptive_packet_header.cc:141-199), chunks_to_symbols, ofdm_carrier_allocator_cvc,
fft_vcc(inverse, shift), ofdm_cyclic_prefixer.  It is not on the accelerated path;
it exists so that benchmarks and examples can produce on-air frames without the test oracle.
"""
import ctypes as C
import zlib

import numpy as np

from . import ofdm_adaptive
from .fec import make_ldpc_encoders
from .ofdm_adaptive_config import get_bits_per_symbol


def constellation_points(cnst):
    """gr::digital constellation_{bpsk,qpsk,8psk,16qam} tables; QPSK scaled by 0.5
    (lib/dtl/constellation.cc:18-24)."""
    cnst = int(cnst)
    if cnst == 1:
        return np.array([-1, 1], dtype=np.complex64)
    if cnst == 2:
        s = np.float32(np.float64(np.float32(0.707107)) * 0.5)
        return np.array([-s - 1j * s, s - 1j * s, -s + 1j * s, s + 1j * s], dtype=np.complex64)
    if cnst == 3:
        a = np.float32(np.pi / 8.0)
        m = np.array([1, 7, 15, 9, 3, 5, 13, 11], dtype=np.float32)
        ang = (m * a).astype(np.float64)         # float product, then cos/sin in double (as the C++ table)
        return (np.cos(ang).astype(np.float32) + 1j * np.sin(ang).astype(np.float32)).astype(np.complex64)
    if cnst == 4:
        level = np.sqrt(np.float32(0.1))
        i = np.arange(16)
        re = np.where(i & 2, 3, 1) * level * np.where(i & 4, -1.0, 1.0)
        im = np.where(i & 1, 3, 1) * level * np.where(i & 8, 1.0, -1.0)
        return (re + 1j * im).astype(np.complex64)
    raise ValueError("Unknown constellation")


def _crc16_header(bits, header_len):
    """pack_crc (packet_header.cc:93-105, including its `i*len+j < header_len` bound) + crc(16, 0x1021,
    0xFFFF, 0, refin=False, refout=True)."""
    ln = (header_len - 16) // 8 + (1 if (header_len - 16) % 8 else 0)
    reg = 0xFFFF
    for i in range(ln):
        byte = 0
        for j in range(8):
            if i * ln + j < header_len:
                byte = ((byte << 1) | int(bits[i * 8 + j] & 1)) & 0xFF
        for b in range(7, -1, -1):
            msb = (reg >> 15) & 1
            reg = (reg << 1) & 0xFFFF
            if msb ^ ((byte >> b) & 1):
                reg ^= 0x1021
    return int("{:016b}".format(reg)[::-1], 2)


def format_header(fields, has_fec, header_len):
    """header_formatter (packet_header.cc:141-199): LSB-first fields + CRC-16, 1 bit per byte."""
    out = np.zeros(header_len, dtype=np.uint8)
    k = 0

    def put(pos, val, nbits):
        for i in range(nbits):
            if pos + i < header_len:
                out[pos + i] = (val >> i) & 1
        return pos + nbits
    k = put(k, fields["payload_len"] & 0xFFF, 12)
    k = put(k, fields["frame_no"] & 0xFFF, 12)
    k = put(k, fields["cnst"] & 0xF, 4)
    k = put(k, fields.get("fb_cnst", 0) & 0xF, 4)
    if has_fec:
        put(k + 0, fields["tb_no"], 12)
        put(k + 12, fields.get("fec_fb", 0), 4)
        put(k + 16, fields["tb_offset"], 12)
        put(k + 28, fields["fec"], 4)
        put(k + 32, fields["tb_payload"], 16)
        k += 48
    put(k, _crc16_header(out, header_len), 16)
    return out


def _bits_lsb_first(data_bytes):
    return np.unpackbits(np.asarray(data_bytes, dtype=np.uint8).reshape(-1, 1), axis=1, bitorder="little").ravel()


def _pack_symbols(bits, bps):
    """repack(1, bps).repack_lsb_first: first bit = LSB of the first symbol; a trailing partial symbol is kept."""
    pad = (-len(bits)) % bps
    b = np.concatenate([bits, np.zeros(pad, np.uint8)]).reshape(-1, bps)
    return (b << np.arange(bps, dtype=np.uint8)).sum(axis=1).astype(np.uint8)


class ofdm_transmitter:
    """Adaptive OFDM Transmitter (python/dtl/ofdm_transmitter.py:11-61), batch form."""

    def __init__(self, config, name="tx"):
        self.name = name
        self._config = config
        self.fft_len, self.cp_len = config.fft_len, config.cp_len
        self.occupied_carriers = config.occupied_carriers
        self.pilot_carriers = config.pilot_carriers
        self.pilot_symbols = config.pilot_symbols
        self.frame_length = config.frame_length
        self.fec = len(config.fec_codes)
        self.codes_alist, self.codes_id = [], {}
        if self.fec:
            self.codes_alist = list(zip(*config.fec_codes))[1]
            self.codes_id = {name: id + 1 for (id, name) in enumerate(list(zip(*config.fec_codes))[0])}
            self.ldpc_encs = make_ldpc_encoders(self.codes_alist)
        if [self.fft_len, self.fft_len] != [len(config.sync_word1), len(config.sync_word2)]:
            raise ValueError("Length of sync sequence(s) must be FFT length.")
        self.sync_words = [np.asarray(config.sync_word1, np.complex64), np.asarray(config.sync_word2, np.complex64)]
        self.header_len = 2 if self.fec else 1
        self.frame_capacity = ofdm_adaptive.frame_capacity(self.frame_length, self.occupied_carriers)
        self.hdr_bits = self.header_len * len(self.occupied_carriers[0])

    @property
    def frame_samples(self):
        return (2 + self.header_len + self.frame_length) * (self.fft_len + self.cp_len)

    # -- framing ----------------------------------------------------------------------------
    def _tb_encode(self, enc, bits, ncws):
        """tb_encoder::encode (lib/dtl/tb_encoder.cc:32-74): [n-k check | k_i data] per codeword."""
        n, k = enc.get_n(), enc.get_k()
        ks, rem = [], len(bits)
        for i in range(ncws):
            kn = -(-rem // (ncws - i))
            ks.append(kn)
            rem -= kn
        data = np.zeros((ncws, k), dtype=np.uint8)
        pos = 0
        for i, kn in enumerate(ks):
            data[i, :kn] = bits[pos:pos + kn]
            pos += kn
        cws = enc.encode_batch(data)
        return np.concatenate([np.concatenate([cws[i, :n - k], cws[i, n - k:n - k + ks[i]]]) for i in range(ncws)])

    def build_fec_frames(self, user_bytes, constellation, fec_scheme, first_frame_no=0, first_tb_no=0):
        """-> (symbols uint8 [n_frames, capacity], [header field dicts])"""
        user = np.asarray(user_bytes, dtype=np.uint8)
        cnst = int(constellation)
        fec_idx = self.codes_id[fec_scheme] if isinstance(fec_scheme, str) else int(fec_scheme)
        enc = self.ldpc_encs[fec_idx]
        bps, cap = get_bits_per_symbol(cnst), self.frame_capacity
        n, k = enc.get_n(), enc.get_k()
        frame_bytes = cap * bps // 8
        ncws = 1 + (frame_bytes * 8) // n if frame_bytes * 8 > n else 1     # compute_tb_len
        tb_payload_max = ncws * k // 8 - 4
        if tb_payload_max <= 0:
            raise RuntimeError("Misconfiguration: TB should be able to carry at least 1 user data bit")
        frames, hdrs = [], []
        cur = np.zeros(cap, dtype=np.uint8)
        st = dict(offset=0, used=0, payload=0, wpos=0, frame_no=first_frame_no, tb_count=first_tb_no, tb_bits=0)

        def emit(payload_bytes):
            nonlocal cur
            hdrs.append(dict(payload_len=payload_bytes, frame_no=st["frame_no"] & 0xFFF, cnst=cnst, fb_cnst=0,
                             tb_no=st["tb_count"] & 0xFF, fec_fb=0, tb_offset=st["offset"] & 0xFFF, fec=fec_idx,
                             tb_payload=st["tb_bits"] & 0xFFFF))
            frames.append(cur)
            cur = np.zeros(cap, dtype=np.uint8)
            st.update(offset=0, used=0, payload=0, wpos=0, frame_no=(st["frame_no"] + 1) & 0xFFF)

        read = 0
        while read < len(user):
            chunk = user[read:read + tb_payload_max]
            crc = zlib.crc32(chunk.tobytes()) & 0xFFFFFFFF
            buf = np.concatenate([chunk, np.frombuffer(crc.to_bytes(4, "little"), np.uint8)])
            tb = self._tb_encode(enc, _bits_lsb_first(buf), ncws)
            st["tb_bits"] = len(buf) * 8
            read += len(chunk)
            st["tb_count"] += 1
            used_frames, idx = 0, 0
            while idx < len(tb):
                remaining = len(tb) - idx
                bytes_left = -(-remaining // 8)
                avail = (cap - st["used"]) * bps // 8
                out_bytes = min(bytes_left, avail)
                nbits = min(out_bytes * 8, remaining)
                syms = _pack_symbols(tb[idx:idx + nbits], bps)
                cur[st["wpos"]:] = 0
                cur[st["wpos"]:st["wpos"] + len(syms)] = syms
                idx += nbits
                if out_bytes == avail:
                    emit(st["used"] * bps // 8 + out_bytes)
                elif st["offset"] == 0:
                    st["offset"] = out_bytes
                    st["payload"] += out_bytes
                    st["used"] = -(-out_bytes * 8 // bps)
                    st["wpos"] += st["used"]
                else:
                    st["used"] = st["offset"] + (-(-out_bytes * 8 // bps))   # sic (:459-460)
                    st["payload"] += out_bytes
                    st["wpos"] += st["used"]
                used_frames += 1
            if (used_frames == 1 or read >= len(user)) and (st["payload"] > 0 or st["used"] > 0):
                emit(st["payload"])
        return np.stack(frames) if frames else np.zeros((0, cap), np.uint8), hdrs

    def build_nofec_frames(self, user_bytes, constellation, first_frame_no=0):
        user = np.asarray(user_bytes, dtype=np.uint8)
        cnst = int(constellation)
        bps, cap = get_bits_per_symbol(cnst), self.frame_capacity
        per = cap * bps // 8 - 4
        frames, hdrs = [], []
        for f, lo in enumerate(range(0, len(user), per)):
            chunk = user[lo:lo + per]
            buf = np.zeros(per + 4, dtype=np.uint8)
            buf[:len(chunk)] = chunk
            crc = zlib.crc32(chunk.tobytes()) & 0xFFFFFFFF
            buf[len(chunk):len(chunk) + 4] = np.frombuffer(crc.to_bytes(4, "little"), np.uint8)
            frames.append(_pack_symbols(_bits_lsb_first(buf), bps)[:cap])
            hdrs.append(dict(payload_len=len(chunk) + 4, frame_no=(first_frame_no + f) & 0xFFF, cnst=cnst, fb_cnst=0))
        return np.stack(frames), hdrs

    # -- OFDM modulation ----------------------------------------------------------------------
    def modulate(self, frame_syms, hdrs):
        """allocator + IFFT(shift) + cyclic prefix -> complex64 [n_frames, frame_samples]"""
        N, cp, h, L = self.fft_len, self.cp_len, self.header_len, self.frame_length
        nfr = len(hdrs)
        grid = np.zeros((nfr, 2 + h + L, N), dtype=np.complex64)
        grid[:, 0, :] = self.sync_words[0]
        grid[:, 1, :] = self.sync_words[1]
        bpsk = constellation_points(1)

        def sh(c):
            return ((c + N if c < 0 else c) + N // 2) % N
        hdr_bits = np.stack([format_header(hd, bool(self.fec), self.hdr_bits) for hd in hdrs])
        hpos = 0
        for i in range(h + L):
            occ = self.occupied_carriers[i % len(self.occupied_carriers)]
            idx = [sh(c) for c in occ]
            if i < h:
                grid[:, 2 + i, idx] = bpsk[hdr_bits[:, hpos:hpos + len(occ)]]
                hpos += len(occ)
            pset = self.pilot_carriers[i % len(self.pilot_carriers)]
            grid[:, 2 + i, [sh(c) for c in pset]] = np.asarray(
                self.pilot_symbols[i % len(self.pilot_symbols)], dtype=np.complex64)
        ppos = 0
        for i in range(L):
            occ = self.occupied_carriers[(h + i) % len(self.occupied_carriers)]
            idx = [sh(c) for c in occ]
            for f in range(nfr):
                pts = constellation_points(hdrs[f]["cnst"])
                grid[f, 2 + h + i, idx] = pts[frame_syms[f, ppos:ppos + len(occ)]]
            ppos += len(occ)
        time = np.fft.ifft(np.fft.ifftshift(grid, axes=-1), axis=-1) * N    # FFTW backward: unnormalised
        out = np.concatenate([time[..., N - cp:], time], axis=-1).astype(np.complex64)
        return out.reshape(nfr, -1)

    def work(self, user_bytes, constellation, fec_scheme=None, first_frame_no=0, first_tb_no=0):
        if self.fec:
            syms, hdrs = self.build_fec_frames(user_bytes, constellation, fec_scheme, first_frame_no, first_tb_no)
        else:
            syms, hdrs = self.build_nofec_frames(user_bytes, constellation, first_frame_no)
        return self.modulate(syms, hdrs), hdrs


    # -- HIP path ---------------------------------------------------------------------------
    def _device_handle(self):
        if getattr(self, "_h", None):
            return self._h
        from . import _lib
        from .ofdm_receiver import make_cfg_struct
        L = _lib.lib()
        if not getattr(L, "_tx_declared", False):
            L.dtl_tx_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
            L.dtl_tx_destroy.argtypes = [C.c_void_p]
            L.dtl_tx_destroy.restype = None
            L.dtl_tx_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_void_p, C.POINTER(_TxResult)]
            L.dtl_copy_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
            L._tx_declared = True
        c, self._keep = make_cfg_struct(self._config)
        h = C.c_void_p()
        _lib.check(L.dtl_tx_create(C.byref(c), C.byref(h)))
        self._h = h
        return h

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                from . import _lib
                _lib.lib().dtl_tx_destroy(self._h)
        except Exception:
            pass

    def plan_device(self, n_user, constellation, fec_scheme=None, first_frame_no=0, first_tb_no=0):
        """Header records of the frames work_device() would emit for n_user bytes (host only: the framing state
        machine of the C++ side evaluated on lengths)."""
        from . import _lib
        h = self._device_handle()
        L = _lib.lib()
        L.dtl_tx_plan.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                  C.POINTER(C.c_int64)]
        fec_idx = 0
        if self.fec:
            fec_idx = self.codes_id[fec_scheme] if isinstance(fec_scheme, str) else int(fec_scheme)
        cap = int(n_user) // 8 + 64
        out = np.zeros(cap, dtype=_TXF_DTYPE)
        n_out = C.c_int64(0)
        _lib.check(L.dtl_tx_plan(h, int(n_user), int(constellation), fec_idx, int(first_frame_no), int(first_tb_no),
                                 out.ctypes.data, cap, C.byref(n_out)))
        return out[:int(n_out.value)].copy()

    def work_device(self, user, constellation, fec_scheme=None, first_frame_no=0, first_tb_no=0, gap=0,
                    want_syms=False):
        """user: uint8 CUDA tensor.  -> (iq complex64 CUDA tensor [n_frames, frame_samples + gap], header records
        (numpy structured array), symbol indices uint8 [n_frames, capacity] or None)"""
        import torch
        from . import _lib
        if not (hasattr(user, "is_cuda") and user.is_cuda):
            raise _lib.DtlError(_lib.ERR_NO_DEVICE, "ofdm_transmitter.work_device needs a device tensor (no CPU fallback)")
        if user.dtype != torch.uint8:
            raise ValueError("user bytes must be uint8")
        user = user.contiguous().view(-1)
        fec_idx = 0
        if self.fec:
            fec_idx = self.codes_id[fec_scheme] if isinstance(fec_scheme, str) else int(fec_scheme)
        res = _TxResult()
        st = _lib.current_stream_ptr()
        h = self._device_handle()
        _lib.check(_lib.lib().dtl_tx_process(h, user.data_ptr(), user.numel(), int(constellation), fec_idx,
                                             int(first_frame_no), int(first_tb_no), int(gap), int(bool(want_syms)), st,
                                             C.byref(res)))
        nf = int(res.n_frames)
        hdrs = np.empty(nf, dtype=_TXF_DTYPE)
        if nf:
            C.memmove(hdrs.ctypes.data, res.frames, nf * hdrs.itemsize)
        dev = user.device

        def grab(ptr, shape, dtype):
            t = torch.empty(shape, dtype=dtype, device=dev)
            if t.numel() and ptr:
                _lib.check(_lib.lib().dtl_copy_device(t.data_ptr(), ptr, t.numel() * t.element_size(), st))
            return t
        iq = grab(res.d_iq, (nf, int(res.frame_samples) + int(gap)), torch.complex64)
        syms = grab(res.d_syms, (nf, int(res.frame_capacity)), torch.uint8) if want_syms else None
        return iq, hdrs, syms


class TxFrameInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("payload_len", "frame_no", "cnst", "fb_cnst", "tb_no", "fec_fb", "tb_offset",
                                         "fec", "tb_payload", "pad")]


class _TxResult(C.Structure):
    _fields_ = [("n_frames", C.c_int64), ("n_samples", C.c_int64), ("n_tbs", C.c_int64),
                ("frames", C.POINTER(TxFrameInfo)), ("d_iq", C.c_void_p), ("d_syms", C.c_void_p),
                ("d_hdr_bits", C.c_void_p), ("frame_capacity", C.c_int32), ("hdr_bits", C.c_int32),
                ("frame_samples", C.c_int32), ("pad", C.c_int32)]


_TXF_DTYPE = np.dtype([(n, "<i4") for n, _ in TxFrameInfo._fields_])


class ofdm_adaptive_tx:
    """python/dtl/ofdm_adaptive_tx.py:10-25"""

    @classmethod
    def from_parameters(cls, config_dict=None, **kwargs):
        from .ofdm_adaptive_config import make_tx_config
        cfg = make_tx_config(config_dict)
        for key, val in kwargs.items():
            if hasattr(cfg, key):
                setattr(cfg, key, val)
        return cls(cfg)

    def __init__(self, config):
        self.config = config
        self.direct_tx = ofdm_transmitter(config)

    def work(self, user_bytes, constellation, fec_scheme=None, first_frame_no=0, first_tb_no=0):
        return self.direct_tx.work(user_bytes, constellation, fec_scheme, first_frame_no, first_tb_no)

    def work_device(self, user, constellation, fec_scheme=None, **kwargs):
        return self.direct_tx.work_device(user, constellation, fec_scheme, **kwargs)
