"""The single-precision math the soft demapper and the oscillator are DEFINED with (oracle/orc_math.h; the kernels execute
the same operation sequences: tools/math_check.hip, and every parity test): how far they are from the true functions.
The reference calls the C library's logf / expf and reads a 10-bit sine table -- none reproducible bit for bit -- so these
definitions are ours, and this is their accuracy claim: the logarithm within 0.85 ulp, the
exponential within 1 ulp on its whole range, the oscillator's sine / cosine within 1.5e-7 absolute."""
import ctypes as C

import numpy as np


def _ulp_err(got, want64):
    want32 = want64.astype(np.float32)
    ulp = np.spacing(np.abs(want32)).astype(np.float64)
    ulp = np.maximum(ulp, 2.0 ** -149)
    return np.abs(got.astype(np.float64) - want64) / ulp


def test_logarithm_accuracy(oracle):
    """fdlibm's e_logf kernel with its operations written out (one IEEE division): every float of four binades around 1,
    every 257th float of the whole positive range and the subnormals' ends.  (Round 4 also built a division-free table form
    -- 0.53 ulp, 20 operations instead of 30 -- whose table reads cost the demapper more than the division they replaced:
    profiles/r04/table_logarithm_experiment.txt.)"""
    L = oracle.lib()
    L.orc_math_logf.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
    worst = 0.0
    for bits in (np.arange(0x3e800000, 0x40800000, dtype=np.uint32),            # [0.25, 4): every float
                 np.arange(1, 0x7f800000, 257, dtype=np.uint32),                 # the whole range, strided
                 np.arange(1, 1 << 16, dtype=np.uint32), np.arange(0x007f0000, 0x00810000, dtype=np.uint32)):
        x = bits.view(np.float32)
        y = np.empty_like(x)
        L.orc_math_logf(x.ctypes.data, x.size, y.ctypes.data)
        worst = max(worst, float(_ulp_err(y, np.log(x.astype(np.float64))).max()))
    assert 0.8 < worst <= 0.85, worst                                            # measured 0.840
    sp = np.array([0.0, -0.0, -1.0, np.inf, np.nan, 1.0], np.float32)
    out = np.empty_like(sp)
    L.orc_math_logf(sp.ctypes.data, sp.size, out.ctypes.data)
    assert out[0] == -np.inf and out[1] == -np.inf and np.isnan(out[2]) and out[3] == np.inf and np.isnan(out[4]) and out[5] == 0.0


def test_exponential_within_one_ulp(oracle):
    L = oracle.lib()
    L.orc_math_expf.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-104, 89, 4_000_000), rng.uniform(-1, 1, 1_000_000), -np.abs(rng.standard_normal(1_000_000)) * 30]).astype(np.float32)
    y = np.empty_like(x)
    L.orc_math_expf(x.ctypes.data, x.size, y.ctypes.data)
    with np.errstate(over="ignore"):
        want = np.exp(x.astype(np.float64))
    ok = want < 3.4e38
    err = _ulp_err(y[ok], want[ok])
    assert err.max() <= 1.0, (float(err.max()), float(x[ok][int(err.argmax())]))


def test_oscillator_sine_cosine(oracle):
    L = oracle.lib()
    L.orc_math_sincos_phase.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(6)
    a = np.concatenate([rng.integers(0, 1 << 32, 4_000_000, dtype=np.uint64).astype(np.uint32),
                        np.array([0, 1, 0x1fffffff, 0x20000000, 0x20000001, 0x3fffffff, 0x40000000, 0x7fffffff, 0x80000000, 0xffffffff], np.uint32)])
    sn, cs = np.empty(a.size, np.float32), np.empty(a.size, np.float32)
    L.orc_math_sincos_phase(a.ctypes.data, a.size, sn.ctypes.data, cs.ctypes.data)
    ang = a.astype(np.int32).astype(np.float64) * (np.pi / 2 ** 31)
    assert np.abs(sn - np.sin(ang)).max() <= 1.5e-7 and np.abs(cs - np.cos(ang)).max() <= 1.5e-7
    assert sn[-10] == 0.0 and cs[-10] == 1.0                    # phase 0: (1, 0) exactly -- the value before the first trigger
