"""Single-stage entry points of the C ABI (dtl_sync_sc, dtl_fft_batch, dtl_soft_demap, dtl_crc32) against the oracle's
restatement of the corresponding reference blocks: integer outputs exact, floats bit-identical (same operation
sequences on both sides)."""
import zlib

import numpy as np
import pytest

from helpers import make_pair, make_stream


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 256, 1024, 128])
def test_fft_batch_matches_oracle(oracle, dtl, cuda, n):
    import torch
    from gr_dtl_amd import stages
    rng = np.random.default_rng(n)
    x = (rng.standard_normal((7, n)) + 1j * rng.standard_normal((7, n))).astype(np.complex64)
    for forward in (True, False):
        got = stages.fft_batch(torch.from_numpy(x).to(cuda), forward=forward).cpu().numpy()
        want = oracle.fft_shifted(x, forward=forward)
        np.testing.assert_array_equal(_bits(got), _bits(want))
    # forward then inverse returns N x the input (unnormalised, as FFTW / fft_vcc)
    xt = torch.from_numpy(x).to(cuda)
    back = stages.fft_batch(stages.fft_batch(xt, True), False).cpu().numpy() / n
    np.testing.assert_allclose(back, x, rtol=0, atol=2e-6 * np.sqrt(n))


@pytest.mark.gpu
@pytest.mark.parametrize("cnst", [1, 2, 3, 4])
@pytest.mark.parametrize("dist_mode", [0, 1])
def test_soft_demap_matches_oracle(oracle, dtl, cuda, cnst, dist_mode):
    import torch
    from gr_dtl_amd import stages
    rng = np.random.default_rng(10 * cnst + dist_mode)
    syms = ((rng.standard_normal(5000) + 1j * rng.standard_normal(5000)) * 0.8).astype(np.complex64)
    for noise in (None, 0.05, 0.5, 1e-6):
        got = stages.soft_demap(torch.from_numpy(syms).to(cuda), cnst, noise, dist_mode).cpu().numpy()
        want = oracle.soft_demap(cnst, syms, noise, dist_mode)
        same = (_bits(got) == _bits(want)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (cnst, dist_mode, noise, int((~same).sum()))


@pytest.mark.gpu
def test_sync_sc_matches_oracle(oracle, dtl, cuda):
    import torch
    from gr_dtl_amd import stages
    for fft_len, thr in ((64, 0.95), (256, 0.98)):
        pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=fft_len, frame_length=10, sync_threshold=thr)
        user, hdrs, x = make_stream(oracle, ocfg, code, 2, 4000, 18.0, seed=fft_len)
        trig, ang, metric = stages.sync_sc(torch.from_numpy(x).to(cuda), fft_len, ocfg.cp_len, thr, want_metric=True)
        t_ref, a_ref, m_ref = oracle.sync_sc(ocfg, x, want_metric=True)
        pos = np.nonzero(t_ref)[0]
        np.testing.assert_array_equal(trig, pos)
        assert len(trig) >= len(hdrs) - 1
        np.testing.assert_allclose(ang, a_ref[pos], rtol=1e-4, atol=2e-6)
        m = metric.cpu().numpy()
        ok = np.isfinite(m_ref) & np.isfinite(m)
        # window sums in f64 on both sides (different summation orders): the float metrics agree to the last bits
        np.testing.assert_allclose(m[ok], m_ref[ok], rtol=2e-6, atol=1e-30)


@pytest.mark.gpu
def test_crc32_matches_zlib(dtl, cuda):
    import torch
    from gr_dtl_amd import stages
    rng = np.random.default_rng(4)
    for n in (0, 1, 3, 63, 64, 65, 988, 4097, 1 << 20):
        data = rng.integers(0, 256, n, dtype=np.uint8)
        assert stages.crc32(torch.from_numpy(data).to(cuda)) == (zlib.crc32(data.tobytes()) & 0xFFFFFFFF), n


def test_stage_entry_points_refuse_host_memory(dtl):
    import torch
    from gr_dtl_amd import stages
    for call in (lambda: stages.fft_batch(torch.zeros(64, dtype=torch.complex64)),
                 lambda: stages.soft_demap(torch.zeros(4, dtype=torch.complex64), 2),
                 lambda: stages.crc32(torch.zeros(4, dtype=torch.uint8)),
                 lambda: stages.sync_sc(torch.zeros(100, dtype=torch.complex64), 64, 16)):
        with pytest.raises(dtl._lib.DtlError):
            call()
