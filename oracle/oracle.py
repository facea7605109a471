"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package (gr_dtl_amd).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("DTL_ORACLE_LIB") or os.path.join(_HERE, "_build", "liboracle.so")  # override: a sanitizer build


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
    L.orc_ldpc_decode_arith.restype = C.c_int
    L.orc_ldpc_decode_arith.argtypes = [_P, _P, C.c_int, C.c_float, C.c_int, _P, _P]
    L.orc_ldpc_decode_batch_arith.argtypes = [_P, _P, C.c_int, C.c_int, C.c_float, C.c_int, _P, _P, C.c_int]
    L.orc_tbdec_set_arith.argtypes = [_P, C.c_int]


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


NO_CARRIER = 0x7fffffff         # oracle.h: ORC_NO_CARRIER
GR_EXACT, FAST = 0, 1           # arithmetic modes (oracle.h: orc_cfg.arith); DEMAP_FAST is the flag on dist_mode
DEMAP_FAST = 0x100


def soft_demap(cnst, syms, noise=None, dist_mode=0, arith=GR_EXACT):
    if arith:
        dist_mode = int(dist_mode) | DEMAP_FAST
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

    def decode(self, llr, max_it=15, sigma=np.sqrt(2.0), arith=GR_EXACT):
        llr = np.ascontiguousarray(llr, dtype=np.float32)
        assert llr.size == self.n
        out = np.zeros(self.k, dtype=np.uint8)
        nit = C.c_int(0)
        with np.errstate(all="ignore"):
            lib().orc_ldpc_decode_arith(self._h, _ptr(llr), max_it, float(np.float32(sigma)), int(arith), _ptr(out),
                                        C.byref(nit))
        return out, nit.value

    def decode_batch(self, llr, max_it=15, sigma=np.sqrt(2.0), n_threads=1, arith=GR_EXACT):
        llr = np.ascontiguousarray(llr, dtype=np.float32).reshape(-1, self.n)
        ncw = llr.shape[0]
        out = np.zeros((ncw, self.k), dtype=np.uint8)
        nit = np.zeros(ncw, dtype=np.int32)
        lib().orc_ldpc_decode_batch_arith(self._h, _ptr(llr), ncw, max_it, float(np.float32(sigma)), int(arith),
                                          _ptr(out), _ptr(nit), n_threads)
        return out, nit


# ---------------------------------------------------------------- structures

class Header(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("payload_len", "frame_no", "cnst", "fb_cnst", "tb_no",
                                       "fec_fb", "tb_offset", "fec", "tb_payload")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Feedback(C.Structure):
    _fields_ = [("n_mcs", C.c_int), ("snr_th", C.c_double * 16), ("cnst", C.c_int * 16),
                ("fec", C.c_int * 16), ("hysteresis", C.c_double), ("decision_th", C.c_int),
                ("decision_counter", C.c_int), ("last_decision", C.c_int), ("new_decision", C.c_int)]


class TbRec(C.Structure):
    _fields_ = [("tb_no", C.c_int), ("tb_payload", C.c_int), ("crc_ok", C.c_int), ("avg_it", C.c_int),
                ("user_bits", C.c_int), ("out_offset", C.c_long)]


class Cfg(C.Structure):
    _fields_ = [("fft_len", C.c_int), ("cp_len", C.c_int), ("frame_length", C.c_int),
                ("header_len", C.c_int), ("has_fec", C.c_int), ("n_occ_sets", C.c_int),
                ("max_occ", C.c_int), ("occ_len", _P), ("occ", _P), ("n_pilot_sets", C.c_int),
                ("n_pilots", C.c_int), ("pilot_carriers", _P), ("n_pilot_sym_sets", C.c_int),
                ("pilot_symbols", _P), ("sync1", _P), ("sync2", _P), ("sync_threshold", C.c_float),
                ("eq_alpha", C.c_float), ("scramble_header", C.c_int), ("llr_dist_mode", C.c_int),
                ("bp_sigma", C.c_float), ("max_it", C.c_int), ("use_sync_correct", C.c_int), ("arith", C.c_int)]


class Snr(C.Structure):
    _fields_ = [("snr_db", C.c_double), ("noise_db", C.c_double), ("noise_lin", C.c_double)]


class FrameRec(C.Structure):
    _fields_ = [("trigger", C.c_long), ("header_ok", C.c_int), ("hdr", Header),
                ("n_payload_syms", C.c_int), ("carr_offset", C.c_int), ("snr_db", C.c_double),
                ("noise_lin", C.c_double), ("fb_cnst", C.c_int), ("fb_fec", C.c_int),
                ("lost_frames", C.c_int)]


class RxOut(C.Structure):
    _fields_ = [("max_frames", C.c_int), ("frames", _P), ("trig_raw", _P), ("trig_fixed", _P),
                ("angle", _P), ("eq_hard", _P), ("eq_soft", _P), ("soft_syms", _P), ("llr", _P),
                ("hdr_bits", _P), ("taps", _P), ("user_bits", _P), ("user_bits_cap", C.c_long),
                ("n_user_bits", C.c_long), ("user_bytes", _P), ("user_bytes_cap", C.c_long),
                ("n_user_bytes", C.c_long), ("tbs", _P), ("tbs_cap", C.c_int), ("n_tbs", C.c_int),
                ("n_frames", C.c_int), ("cw_nit", _P), ("cw_nit_cap", C.c_long), ("n_cw_nit", C.c_long)]


class FrameDetect(C.Structure):
    _fields_ = [("frame_len", C.c_int), ("remainder", C.c_int), ("missing_count", C.c_long),
                ("correction_count", C.c_long), ("error_count", C.c_long), ("acc_error", C.c_int),
                ("trigger_counter", C.c_int), ("synced", C.c_int), ("in_sync", C.c_int)]


# ---------------------------------------------------------------- header / feedback

def header_format(fields, has_fec, header_len, scramble=False):
    h = Header(**fields)
    out = np.zeros(header_len, dtype=np.uint8)
    lib().orc_header_format(C.byref(h), int(has_fec), header_len, int(scramble), _ptr(out))
    return out


def header_parse(bits, has_fec, state_cnst=QAM16, scramble=False):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    h = Header()
    st = C.c_int(state_cnst)
    ns = C.c_int(0)
    L = lib()
    L.orc_header_parse.restype = C.c_int
    ok = L.orc_header_parse(_ptr(bits), int(has_fec), bits.size, int(scramble), C.byref(st),
                            C.byref(h), C.byref(ns))
    return bool(ok), h.as_dict(), ns.value, st.value


class FeedbackDecision:
    """ofdm_adaptive_feedback_decision(hysteresis, decision_th, lut, initial_mcs)"""

    def __init__(self, hysteresis, decision_th, lut, initial_mcs=0):
        self.s = Feedback()
        n = len(lut)
        th = (C.c_double * n)(*[float(x[0]) for x in lut])
        cn = (C.c_int * n)(*[int(x[1][0]) for x in lut])
        fe = (C.c_int * n)(*[int(x[1][1]) for x in lut])
        lib().orc_feedback_init(C.byref(self.s), C.c_double(hysteresis), int(decision_th), n, th, cn, fe,
                                int(initial_mcs))

    def get_feedback(self, snr):
        c, f = C.c_int(0), C.c_int(0)
        lib().orc_feedback_get(C.byref(self.s), C.c_double(snr), C.byref(c), C.byref(f))
        return c.value, f.value


# ---------------------------------------------------------------- OFDM configuration

PILOT_SCRAMBLE_SEQ = (
    1, 1, 1, 1, -1, -1, -1, 1, -1, -1, -1, -1, 1, 1, -1, 1, -1, -1, 1, 1, -1, 1, 1, -1, 1, 1, 1, 1, 1, 1, -1, 1,
    1, 1, -1, 1, 1, -1, -1, 1, 1, 1, -1, 1, -1, -1, -1, 1, -1, 1, -1, -1, 1, -1, -1, 1, 1, 1, 1, 1, -1, -1, 1, 1,
    -1, -1, 1, -1, 1, -1, 1, 1, -1, -1, -1, 1, 1, -1, -1, -1, -1, 1, -1, -1, 1, -1, 1, 1, 1, 1, -1, 1, -1, 1, -1, 1,
    -1, -1, -1, -1, -1, 1, -1, 1, 1, -1, 1, -1, 1, 1, 1, -1, -1, 1, -1, -1, -1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1)


def make_sync_words(fft_len, occupied_carriers, pilot_carriers):
    """GR3.10 digital.ofdm_txrx._make_sync_word1/2 (python/dtl/ofdm_adaptive_config.py:33-36)."""
    active = [c + fft_len if c < 0 else c for c in list(occupied_carriers[0]) + list(pilot_carriers[0])]
    active = set(active)
    rs = np.random.RandomState(42)
    s2 = np.sqrt(2)
    sw1 = [({0: s2, 1: -s2}[rs.randint(2)] if (x in active and x % 2) else 0) for x in range(fft_len)]
    rs = np.random.RandomState(42)
    sw2 = [({0: 1, 1: -1}[rs.randint(2)] if x in active else 0) for x in range(fft_len)]
    sw2[0] = 0
    return (np.fft.fftshift(np.array(sw1)).astype(c32), np.fft.fftshift(np.array(sw2)).astype(c32))


def default_carriers(fft_len=64):
    """python/dtl/ofdm_adaptive_config.py:17-32 for fft_len 64; for 64*R the 64-map is widened
    R-fold (base carrier b -> carriers b*R .. b*R +/- (R-1)), SURVEY.md 8a/8d."""
    base_occ = (list(range(-26, -21)) + list(range(-20, -7)) + list(range(-6, 0)) + list(range(1, 7))
                + list(range(8, 21)) + list(range(22, 27)))
    base_pil = (-21, -7, 7, 21)
    R = fft_len // 64
    assert fft_len == 64 * R and R >= 1

    def widen(b):
        return [b * R + j if b > 0 else b * R - j for j in range(R)]
    occ = sorted(c for b in base_occ for c in widen(b))
    pil = sorted(c for b in base_pil for c in widen(b))
    pat = {b: s for b, s in zip(base_pil, (1, 1, 1, -1))}
    pil_sign = [pat[int(np.sign(c) * (abs(c) // R))] for c in pil]
    pilot_symbols = [[x * s for s in pil_sign] for x in PILOT_SCRAMBLE_SEQ]
    return (occ,), (tuple(pil), tuple(pil)), pilot_symbols


class Config:
    """Plain-data OFDM configuration + the ctypes orc_cfg view of it."""

    def __init__(self, fft_len=64, cp_len=None, frame_length=20, has_fec=True, sync_threshold=0.95,
                 eq_alpha=0.1, llr_dist_mode=0, bp_sigma=float(np.sqrt(np.float32(2.0))), max_it=15,
                 use_sync_correct=True, occupied_carriers=None, pilot_carriers=None, pilot_symbols=None,
                 sync_word1=None, sync_word2=None, scramble_header=False, arith=GR_EXACT):
        self.fft_len = fft_len
        self.cp_len = fft_len // 4 if cp_len is None else cp_len
        self.frame_length = frame_length
        self.has_fec = bool(has_fec)
        self.header_len = 2 if has_fec else 1   # ofdm_receiver.py:66-68
        if occupied_carriers is None or pilot_carriers is None or pilot_symbols is None:
            occ, pil, psym = default_carriers(fft_len)
        else:
            occ = pil = psym = None
        self.occupied_carriers = occupied_carriers if occupied_carriers is not None else occ
        self.pilot_carriers = pilot_carriers if pilot_carriers is not None else pil
        self.pilot_symbols = pilot_symbols if pilot_symbols is not None else psym
        if sync_word1 is None or sync_word2 is None:
            sync_word1, sync_word2 = make_sync_words(fft_len, self.occupied_carriers, self.pilot_carriers)
        self.sync_word1 = np.ascontiguousarray(sync_word1, dtype=c32)
        self.sync_word2 = np.ascontiguousarray(sync_word2, dtype=c32)
        self.sync_threshold, self.eq_alpha = sync_threshold, eq_alpha
        self.llr_dist_mode, self.bp_sigma, self.max_it = llr_dist_mode, bp_sigma, max_it
        self.use_sync_correct, self.scramble_header = use_sync_correct, scramble_header
        self.arith = int(arith)
        self._build()

    def _build(self):
        nocc = len(self.occupied_carriers)
        self._occ_len = np.array([len(o) for o in self.occupied_carriers], dtype=np.int32)
        mo = int(self._occ_len.max())
        self._occ = np.zeros((nocc, mo), dtype=np.int32)
        for i, o in enumerate(self.occupied_carriers):
            self._occ[i, :len(o)] = o
        # pilot sets may differ in size (the reference's equaliser QA uses empty ones): padded with NO_CARRIER / 0
        npil = max([len(p) for p in self.pilot_carriers] + [1])
        self._pil = np.full((len(self.pilot_carriers), npil), NO_CARRIER, dtype=np.int32)
        for i, pset in enumerate(self.pilot_carriers):
            self._pil[i, :len(pset)] = list(pset)
        self._psym = np.zeros((len(self.pilot_symbols), npil), dtype=c32)
        for i, row in enumerate(self.pilot_symbols):
            self._psym[i, :len(row)] = list(row)
        c = Cfg()
        c.fft_len, c.cp_len, c.frame_length = self.fft_len, self.cp_len, self.frame_length
        c.header_len, c.has_fec = self.header_len, int(self.has_fec)
        c.n_occ_sets, c.max_occ = nocc, mo
        c.occ_len, c.occ = _ptr(self._occ_len).value, _ptr(self._occ).value
        c.n_pilot_sets, c.n_pilots = self._pil.shape
        c.pilot_carriers = _ptr(self._pil).value
        c.n_pilot_sym_sets = self._psym.shape[0]
        c.pilot_symbols = _ptr(self._psym).value
        c.sync1, c.sync2 = _ptr(self.sync_word1).value, _ptr(self.sync_word2).value
        c.sync_threshold, c.eq_alpha = self.sync_threshold, self.eq_alpha
        c.scramble_header, c.llr_dist_mode = int(self.scramble_header), int(self.llr_dist_mode)
        c.bp_sigma, c.max_it, c.use_sync_correct = self.bp_sigma, self.max_it, int(self.use_sync_correct)
        c.arith = int(self.arith)
        self.c = c

    @property
    def frame_capacity(self):
        return lib().orc_frame_capacity(C.byref(self.c))

    @property
    def n_data(self):
        return int(self._occ_len[0])

    @property
    def symbol_len(self):
        return self.fft_len + self.cp_len

    @property
    def frame_samples(self):
        return (2 + self.header_len + self.frame_length) * self.symbol_len


# ---------------------------------------------------------------- stage wrappers

def fft_shifted(x, forward=True):
    x = np.ascontiguousarray(x, dtype=c32)
    out = np.zeros_like(x)
    for i in range(x.reshape(-1, x.shape[-1]).shape[0]):
        lib().orc_fft_shifted(_ptr(x.reshape(-1, x.shape[-1])[i]), _ptr(out.reshape(-1, x.shape[-1])[i]),
                              x.shape[-1], int(forward))
    return out


def tb_encode(code, bits, ncws):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    out = np.zeros(ncws * code.n + 64, dtype=np.uint8)
    L = lib()
    L.orc_tb_encode.restype = C.c_int
    n = L.orc_tb_encode(C.c_void_p(code._h), _ptr(bits), bits.size, ncws, _ptr(out))
    return out[:n].copy()


def fec_frames_build(code, fec_idx, cnst, frame_capacity, user_bytes, first_frame_no=0):
    """-> (symbols uint8 [n_frames, frame_capacity], [header dicts])"""
    user = np.ascontiguousarray(user_bytes, dtype=np.uint8)
    bps = bits_per_symbol(cnst)
    frame_bytes = frame_capacity * bps // 8
    ncws = 1 + (frame_bytes * 8) // code.n if frame_bytes * 8 > code.n else 1
    tb_payload_max = ncws * code.k // 8 - 4
    n_tbs = -(-user.size // max(tb_payload_max, 1))
    tb_bytes = -(-(ncws * code.n) // 8)
    max_frames = n_tbs * (-(-tb_bytes // max(frame_bytes, 1)) + 1) + 4
    syms = np.zeros((max_frames, frame_capacity), dtype=np.uint8)
    hdrs = (Header * max_frames)()
    L = lib()
    L.orc_fec_frames_build.restype = C.c_int
    n = L.orc_fec_frames_build(C.c_void_p(code._h), int(fec_idx), int(cnst), frame_capacity, _ptr(user),
                               user.size, first_frame_no, _ptr(syms), hdrs, max_frames)
    if n < 0:
        raise RuntimeError("orc_fec_frames_build failed: %d" % n)
    return syms[:n].copy(), [hdrs[i].as_dict() for i in range(n)]


def nofec_frames_build(cnst, frame_capacity, user_bytes, first_frame_no=0):
    user = np.ascontiguousarray(user_bytes, dtype=np.uint8)
    bps = bits_per_symbol(cnst)
    per = frame_capacity * bps // 8 - 4
    nfr = (user.size + per - 1) // per
    syms = np.zeros((nfr, frame_capacity), dtype=np.uint8)
    hdrs = []
    L = lib()
    L.orc_nofec_frame_build.restype = C.c_int
    for i in range(nfr):
        chunk = np.ascontiguousarray(user[i * per:(i + 1) * per])
        h = Header()
        L.orc_nofec_frame_build(int(cnst), frame_capacity, _ptr(chunk), chunk.size,
                                (first_frame_no + i) & 0xFFF, _ptr(syms[i]), C.byref(h))
        hdrs.append(h.as_dict())
    return syms, hdrs


def tx_frames(cfg, frame_syms, hdrs, cnst_per_frame=None):
    """OFDM-modulate frames -> complex64 [n_frames, frame_samples]."""
    nfr = frame_syms.shape[0]
    hdr_len = cfg.header_len * cfg.n_data
    out = np.zeros((nfr, cfg.frame_samples), dtype=c32)
    for i in range(nfr):
        bits = header_format(hdrs[i], cfg.has_fec, hdr_len, cfg.scramble_header)
        cn = hdrs[i]["cnst"] if cnst_per_frame is None else cnst_per_frame[i]
        fs = np.ascontiguousarray(frame_syms[i], dtype=np.uint8)
        lib().orc_tx_frame(C.byref(cfg.c), _ptr(bits), _ptr(fs), int(cn), _ptr(out[i]))
    return out


def sync_sc(cfg, iq, want_metric=False):
    iq = np.ascontiguousarray(iq, dtype=c32)
    n = iq.size
    trig = np.zeros(n, dtype=np.uint8)
    angle = np.zeros(n, dtype=np.float32)
    metric = np.zeros(n, dtype=np.float32) if want_metric else None
    with np.errstate(all="ignore"):
        lib().orc_sync_sc(C.byref(cfg.c), _ptr(iq), C.c_long(n), _ptr(metric) if want_metric else None,
                          _ptr(trig), _ptr(angle))
    return (trig, angle, metric) if want_metric else (trig, angle)


def fix_sync(frame_len, trig, chunk=None):
    trig = np.ascontiguousarray(trig, dtype=np.uint8)
    st = FrameDetect()
    lib().orc_framedetect_init(C.byref(st), int(frame_len))
    out = np.zeros_like(trig)
    chunk = trig.size if chunk is None else chunk
    for lo in range(0, trig.size, chunk):
        seg = np.ascontiguousarray(trig[lo:lo + chunk])
        o = np.zeros_like(seg)
        lib().orc_fix_sync(C.byref(st), _ptr(seg), _ptr(o), seg.size)
        out[lo:lo + seg.size] = o
    return out, st


def cfo_mix(cfg, iq, angle, trig=None):
    """trig: the raw trigger flags (uint8 per sample): an oscillator segment starts at every trigger; without them, where
    the held angle changes."""
    iq = np.ascontiguousarray(iq, dtype=c32)
    angle = np.ascontiguousarray(angle, dtype=np.float32)
    out = np.zeros_like(iq)
    t = None if trig is None else np.ascontiguousarray(trig, dtype=np.uint8)
    lib().orc_cfo_mix_trig(C.byref(cfg.c), _ptr(iq), C.c_long(iq.size), _ptr(angle), None if t is None else _ptr(t), _ptr(out))
    return out


def chanest(cfg, sync_syms):
    s = np.ascontiguousarray(sync_syms, dtype=c32)
    taps = np.zeros(cfg.fft_len, dtype=c32)
    L = lib()
    L.orc_chanest.restype = C.c_int
    off = L.orc_chanest(C.byref(cfg.c), _ptr(s), _ptr(taps))
    return off, taps


def frame_equalize(cfg, is_header, cnst, frame, carr_offset, taps):
    frame = np.ascontiguousarray(frame, dtype=c32).reshape(-1, cfg.fft_len)
    n_sym = frame.shape[0]
    taps = np.ascontiguousarray(taps, dtype=c32).copy()
    hard = np.zeros_like(frame)
    soft = np.zeros_like(frame)
    snr = Snr()
    with np.errstate(all="ignore"):
        lib().orc_frame_equalize(C.byref(cfg.c), int(is_header), int(cnst), _ptr(frame), n_sym,
                                 int(carr_offset), _ptr(taps), _ptr(hard), _ptr(soft), C.byref(snr))
    return hard, soft, taps, (snr.snr_db, snr.noise_db, snr.noise_lin)


def serialize(cfg, frame, symbols_skipped, packet_len):
    frame = np.ascontiguousarray(frame, dtype=c32).reshape(-1, cfg.fft_len)
    out = np.zeros(frame.shape[0] * cfg._occ.shape[1] + 8, dtype=c32)
    L = lib()
    L.orc_serialize.restype = C.c_int
    n = L.orc_serialize(C.byref(cfg.c), _ptr(frame), frame.shape[0], int(symbols_skipped), int(packet_len),
                        _ptr(out))
    return out[:n].copy()


class TbDecoder:
    def __init__(self, code, frame_capacity, max_bps=4, max_it=15, bp_sigma=float(np.sqrt(np.float32(2.0))),
                 arith=GR_EXACT):
        L = lib()
        L.orc_tbdec_new.restype = _P
        self._h = C.c_void_p(L.orc_tbdec_new(C.c_void_p(code._h), frame_capacity, max_bps, max_it,
                                              C.c_float(bp_sigma)))
        L.orc_tbdec_set_arith(self._h, int(arith))
        self.code = code

    def __del__(self):
        try:
            lib().orc_tbdec_free(self._h)
        except Exception:
            pass

    def frame(self, llr, cnst, hdr):
        llr = np.ascontiguousarray(llr, dtype=np.float32)
        h = Header(**hdr)
        out = np.zeros(4 * llr.size + 70000, dtype=np.uint8)
        n_out = C.c_long(0)
        recs = (TbRec * 4)()
        nrec = C.c_int(0)
        L = lib()
        L.orc_tbdec_frame.restype = C.c_int
        rc = L.orc_tbdec_frame(self._h, _ptr(llr), llr.size, int(cnst), C.byref(h), _ptr(out),
                               C.c_long(out.size), C.byref(n_out), recs, 4, C.byref(nrec))
        if rc < 0:
            raise RuntimeError("orc_tbdec_frame: %d" % rc)
        return out[:n_out.value].copy(), [{k: getattr(recs[i], k) for k, _ in TbRec._fields_}
                                          for i in range(nrec.value)]


def rx_stream(cfg, code, iq, feedback=None, max_frames=None, want=("frames",)):
    """Run the whole RX graph.  want: any of eq_hard, eq_soft, soft_syms, llr, hdr_bits, taps,
    trig, angle, cw (BP iterations of every codeword, decode order).  Returns a dict."""
    iq = np.ascontiguousarray(iq, dtype=c32)
    n = iq.size
    if max_frames is None:
        max_frames = n // cfg.frame_samples + 16
    if feedback is None:
        feedback = FeedbackDecision(1, 5, [(0.0, (BPSK, 0)), (13, (QPSK, 0)), (18, (PSK8, 0)), (23, (QAM16, 0))], 0)
    L_, N, cap = cfg.frame_length, cfg.fft_len, cfg.frame_capacity
    o = RxOut()
    o.max_frames = max_frames
    keep = {}

    def alloc(name, shape, dtype):
        a = np.zeros(shape, dtype=dtype)
        keep[name] = a
        setattr(o, name, _ptr(a).value)
        return a
    frames = (FrameRec * max_frames)()
    o.frames = C.cast(frames, _P).value
    if "trig" in want:
        alloc("trig_raw", n, np.uint8); alloc("trig_fixed", n, np.uint8)
    if "angle" in want:
        alloc("angle", n, np.float32)
    if "eq_hard" in want:
        alloc("eq_hard", (max_frames, L_, N), c32)
    if "eq_soft" in want:
        alloc("eq_soft", (max_frames, L_, N), c32)
    if "soft_syms" in want:
        alloc("soft_syms", (max_frames, cap), c32)
    if "llr" in want:
        alloc("llr", (max_frames, cap * 4), np.float32)
    if "hdr_bits" in want:
        alloc("hdr_bits", (max_frames, cfg.header_len * cfg.n_data), np.uint8)
    if "taps" in want:
        alloc("taps", (max_frames, N), c32)
    if "cw" in want and code is not None:
        cwn = alloc("cw_nit", max_frames * (cap * 4 // code.n + 2) + 8, np.int32)
        o.cw_nit_cap = cwn.size
    ub = alloc("user_bits", max_frames * cap * 4 + 64, np.uint8)
    o.user_bits_cap = ub.size
    uby = alloc("user_bytes", max_frames * cap // 2 + 64, np.uint8)
    o.user_bytes_cap = uby.size
    tbs = (TbRec * (2 * max_frames + 8))()
    o.tbs = C.cast(tbs, _P).value
    o.tbs_cap = 2 * max_frames + 8
    L = lib()
    L.orc_rx_stream.restype = C.c_int
    with np.errstate(all="ignore"):
        rc = L.orc_rx_stream(C.byref(cfg.c), C.c_void_p(code._h) if code is not None else None,
                             C.byref(feedback.s), _ptr(iq), C.c_long(n), C.byref(o))
    if rc != 0:
        raise RuntimeError("orc_rx_stream: %d" % rc)
    nf = o.n_frames
    res = {"n_frames": nf, "rc": rc}
    recs = []
    for i in range(nf):
        f = frames[i]
        d = {k: getattr(f, k) for k, _ in FrameRec._fields_ if k != "hdr"}
        d["hdr"] = f.hdr.as_dict()
        recs.append(d)
    res["frames"] = recs
    for k, a in keep.items():
        if k in ("user_bits", "user_bytes", "cw_nit"):
            continue
        res[k] = a[:nf] if a.shape[0] == max_frames else a
    if "cw_nit" in keep:
        res["cw_nit"] = keep["cw_nit"][:min(o.n_cw_nit, o.cw_nit_cap)].copy()
    res["user_bits"] = ub[:o.n_user_bits].copy()
    res["user_bytes"] = uby[:o.n_user_bytes].copy()
    ntb = min(o.n_tbs, o.tbs_cap)
    res["tbs"] = [{k: getattr(tbs[i], k) for k, _ in TbRec._fields_} for i in range(ntb)]
    return res


def awgn_stream(frames_iq, noise_var, seed, lead_zeros=100, tail_zeros=None):
    """frames back to back, `lead_zeros` zeros in front (qa_ofdm_adaptive_txrx.py:82-83), complex AWGN of
    total variance noise_var per sample."""
    frames_iq = np.asarray(frames_iq, dtype=c32)
    x = frames_iq.reshape(-1)
    if tail_zeros is None:
        tail_zeros = frames_iq.shape[-1]
    x = np.concatenate([np.zeros(lead_zeros, c32), x, np.zeros(tail_zeros, c32)])
    if noise_var > 0:
        rng = np.random.default_rng(seed)
        s = np.sqrt(noise_var / 2)
        x = x + (rng.normal(0, s, x.size) + 1j * rng.normal(0, s, x.size)).astype(c32)
    return x.astype(c32)
