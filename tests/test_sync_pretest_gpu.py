"""The sync kernel decides most outputs with a single-precision pre-test and leaves the rest to the exact test (double
window sums, exact threshold comparison; rx_kernels.hip pass C).  dtl_sync_pretest_check runs BOTH on every strip: the
pre-test must never decide an output differently from the exact test, on ordinary streams and on inputs built to stress
its error bound (levels grazing the threshold, bursts followed by weak signals, huge / tiny amplitudes, NaN / inf,
zeros)."""
import ctypes as C

import numpy as np
import pytest

from helpers import make_pair, make_stream


def _check(dtl, x, fft_len, thr):
    import torch
    L = dtl._lib.lib()
    L.dtl_sync_pretest_check.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.POINTER(C.c_uint64), C.c_void_p]
    xt = torch.from_numpy(np.ascontiguousarray(x, dtype=np.complex64)).cuda()
    out = (C.c_uint64 * 3)()
    dtl._lib.check(L.dtl_sync_pretest_check(xt.data_ptr(), xt.numel(), fft_len, C.c_float(thr), out, None))
    return int(out[0]), int(out[1]), int(out[2])


def _plateau_stream(rng, N, n_frames, level, jitter):
    """repeated half-symbols (a Schmidl-Cox plateau) whose metric level sits at `level` by construction: the second half
    is the first plus independent noise of the right power; `jitter` moves the level from frame to frame"""
    parts = []
    for _ in range(n_frames):
        lv = min(0.999, max(0.01, level + rng.normal(0, jitter)))
        rho = np.sqrt(lv)                      # |P| / E = rho for unit-power halves with correlation rho
        a = (rng.normal(0, 1, N) + 1j * rng.normal(0, 1, N)) / np.sqrt(2)
        w = (rng.normal(0, 1, N) + 1j * rng.normal(0, 1, N)) / np.sqrt(2)
        b = rho * a + np.sqrt(max(0.0, 1 - rho * rho)) * w
        parts += [a[:N // 2], b[:N // 2], a[N // 2:], b[N // 2:], (rng.normal(0, 1, 3 * N) + 1j * rng.normal(0, 1, 3 * N)) * 0.7]
    return np.concatenate(parts).astype(np.complex64)


@pytest.mark.gpu
def test_pretest_agrees_on_receive_streams(oracle, dtl, cuda):
    pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=256, frame_length=12)
    total = undecided = 0
    for esn0, thr in ((None, 0.95), (30.0, 0.95), (23.0, 0.95), (18.0, 0.95), (14.0, 0.9), (18.0, 0.5)):
        user, hdrs, x = make_stream(oracle, ocfg, code, 4, 40000, esn0, seed=5)
        strips, slow, bad = _check(dtl, x, 256, thr)
        assert bad == 0, (esn0, thr, bad)
        total += strips
        undecided += slow
    assert total > 100000
    assert undecided < 0.25 * total          # zeros in front / behind and the few outputs near the threshold


@pytest.mark.gpu
@pytest.mark.parametrize("fft_len", [64, 256, 1024])
def test_pretest_agrees_on_adversarial_streams(dtl, cuda, fft_len):
    rng = np.random.default_rng(fft_len)
    N = fft_len
    bad_total = strips_total = 0
    cases = []
    for thr in (0.95, 0.9, 0.5, 0.05, 0.999):
        # plateaus whose level grazes the threshold from both sides, with tiny frame-to-frame jitter
        for jit in (0.0, 1e-4, 1e-3, 1e-2):
            cases.append((thr, _plateau_stream(rng, N, 60, thr, jit)))
        # pure noise (the metric wanders around N/2 / N ... far below most thresholds, near the small ones)
        cases.append((thr, (rng.normal(0, 1, 80 * N) + 1j * rng.normal(0, 1, 80 * N)).astype(np.complex64)))
    base = _plateau_stream(rng, N, 40, 0.96, 5e-3)
    for scale in (1e-18, 1e-12, 1e-6, 1.0, 1e6, 1e12, 1e17, 3e18):
        cases.append((0.95, (base * np.complex64(scale)).astype(np.complex64)))
    # a strong burst, then a signal 80 .. 150 dB weaker inside the same 4096-sample tiles
    for drop in (1e-4, 1e-6, 1e-7, 3e-8):
        y = base.copy()
        y[y.size // 3:] *= np.complex64(drop)
        y[: y.size // 3] *= np.complex64(30.0)
        cases.append((0.95, y))
    # zeros, denormals, NaN / inf sprinkled in
    y = base.copy()
    y[1000:1000 + 3 * N] = 0
    y[5000] = np.nan
    y[9000] = np.inf
    y[12000:12010] = 1e-42
    cases.append((0.95, y))
    cases.append((0.95, np.zeros(20 * N, np.complex64)))
    for thr, x in cases:
        strips, slow, bad = _check(dtl, x, N, thr)
        strips_total += strips
        bad_total += bad
        assert bad == 0, (thr, strips, slow, bad)
    assert strips_total > 10000 and bad_total == 0
