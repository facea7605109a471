"""The production ("fast") arithmetic of the decoder and the 16-QAM demapper (include/dtl_rx.h, enum dtl_arith):
the HIP kernels must equal the oracle's restatement of THAT arithmetic bit for bit (orc_ldpc_decode_fast,
ORC_DEMAP_FAST) -- hard decisions, iteration counts, soft bits -- exactly as the default mode equals the oracle's
restatement of the reference's; and the two arithmetics must agree with each other to within the tolerances
north_star states (soft bits 1e-5 relative) / decode the same data."""
import numpy as np
import pytest

from conftest import ALIST_100_23, ALIST_100_27, ALIST_300
from helpers import make_pair, make_stream
from test_ldpc_gpu import _noisy_llrs
from test_rx_gpu import _compare, _run_both


@pytest.mark.gpu
@pytest.mark.parametrize("alist,esn0", [(ALIST_300, -1.0), (ALIST_300, 0.0), (ALIST_300, 1.5), (ALIST_300, 4.0),
                                        (ALIST_100_23, -3.0), (ALIST_100_27, -2.0)])
def test_fast_decoder_matches_fast_oracle(oracle, dtl, cuda, alist, esn0):
    import torch
    code = oracle.Ldpc(alist)
    dec = dtl.ldpc_dec(alist, arith="fast")
    data, llr = _noisy_llrs(code, 3000, esn0, seed=int(esn0 * 10) + code.n + 1)
    ref_bits, ref_nit = code.decode_batch(llr, n_threads=8, arith=oracle.FAST)
    bits, nit = dec.decode_batch(torch.from_numpy(llr).to(cuda))
    bits, nit = bits.cpu().numpy(), nit.cpu().numpy()
    assert 0 < ref_nit.mean() < 15
    np.testing.assert_array_equal(nit, ref_nit)
    np.testing.assert_array_equal(bits, ref_bits)
    # the same handle switched back decodes with the reference's operations again
    dec.set_arith("gr_exact")
    ex_bits, ex_nit = code.decode_batch(llr, n_threads=8)
    bits2, nit2 = dec.decode_batch(torch.from_numpy(llr).to(cuda))
    np.testing.assert_array_equal(nit2.cpu().numpy(), ex_nit)
    np.testing.assert_array_equal(bits2.cpu().numpy(), ex_bits)
    # both arithmetics are the sum-product algorithm: what the fast one decodes wrongly, the exact one does too
    # (the reverse does not hold -- the reference's form collapses to NaN on saturated products)
    fast_err = (ref_bits != data).any(1)
    exact_err = (ex_bits != data).any(1)
    assert fast_err.sum() <= exact_err.sum() + 3


@pytest.mark.gpu
def test_fast_decoder_saturated_and_shortened(oracle, dtl, cuda):
    """+-inf / huge / NaN soft bits and shortened codewords through the fast arithmetic: every intermediate stays
    finite by construction, results equal the oracle's."""
    import torch
    code = oracle.Ldpc(ALIST_300)
    dec = dtl.ldpc_dec(ALIST_300, arith="fast")
    rng = np.random.default_rng(5)
    data = rng.integers(0, 2, (64, code.k)).astype(np.uint8)
    cws = np.stack([code.encode(d) for d in data]).astype(np.float32)
    for scale in (1.0, 40.0, 800.0, np.inf):
        llr = np.where(cws > 0, scale, -scale).astype(np.float32)
        llr2 = llr.copy()
        llr2[::2, 5] *= -1
        llr3 = llr2.copy()
        llr3[1::4, 17] = np.nan
        llr3[::3, 40:60] = 0.0
        for x in (llr, llr2, llr3):
            ref_bits, ref_nit = code.decode_batch(x, arith=oracle.FAST)
            bits, nit = dec.decode_batch(torch.from_numpy(x).to(cuda))
            np.testing.assert_array_equal(nit.cpu().numpy(), ref_nit)
            np.testing.assert_array_equal(bits.cpu().numpy(), ref_bits)
    # a flipped bit among saturated ones is corrected (the exact form turns it into NaN and decides 0)
    llr = np.where(cws > 0, 800.0, -800.0).astype(np.float32)
    llr[:, 5] *= -1
    bits, nit = dec.decode_batch(torch.from_numpy(llr).to(cuda))
    np.testing.assert_array_equal(bits.cpu().numpy(), data)


@pytest.mark.gpu
@pytest.mark.parametrize("noise", [0.5, 0.05, 0.004, None])
def test_fast_demapper_matches_fast_oracle(oracle, dtl, cuda, noise):
    import torch
    from gr_dtl_amd import stages
    rng = np.random.default_rng(21)
    for cnst in (1, 2, 3, 4):
        pts = oracle.constellation_points(cnst)
        sd = np.sqrt((noise if noise is not None else 0.01) / 2)
        s = (pts[rng.integers(0, len(pts), 20000)] + rng.normal(0, sd, 20000) + 1j * rng.normal(0, sd, 20000)).astype(np.complex64)
        ref = oracle.soft_demap(cnst, s, noise, arith=oracle.FAST)
        got = stages.soft_demap(torch.from_numpy(s).to(cuda), cnst, noise, arith="fast").cpu().numpy()
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), cnst
        exact = oracle.soft_demap(cnst, s, noise)
        if cnst != 4:
            assert np.array_equal(ref.view(np.uint32), exact.view(np.uint32))       # only 16-QAM has a fast form
        else:
            fin = np.isfinite(exact) & np.isfinite(ref)
            assert fin.mean() > 0.5 or noise is None or noise < 0.01
            # 1e-5 relative (north_star); soft bits near zero are differences of two logarithms of nearly equal sums
            # and carry the absolute rounding error of those (a few float ulps of the log values)
            err = np.abs(ref[fin] - exact[fin])
            assert (err <= 1e-5 * np.abs(exact[fin]) + 4e-6).all(), float((err / np.maximum(np.abs(exact[fin]), 1e-30)).max())


@pytest.mark.gpu
@pytest.mark.parametrize("cnst,esn0,nbytes", [(2, 12.0, 6000), (3, 20.0, 6000), (4, 21.0, 8000), (4, 24.0, 8000)])
def test_fast_chain_matches_fast_oracle(oracle, dtl, cuda, cnst, esn0, nbytes):
    """Whole receive chain with arith = fast against the oracle chain with arith = FAST: every stage compared as in
    test_rx_gpu (the stages in front of the demapper do not depend on the mode)."""
    pcfg, ocfg, code = make_pair(oracle, dtl, arith="fast")
    user, hdrs, x = make_stream(oracle, ocfg, code, cnst, nbytes, esn0, seed=500 + cnst)
    rx, res, ref = _run_both(oracle, dtl, cuda, pcfg, ocfg, code, x)
    assert ref["n_frames"] >= 0.99 * len(hdrs)
    _compare(rx, res, ref, ocfg, True)
    assert any(t["avg_it"] > 0 for t in ref["tbs"]) or esn0 >= 24.0


@pytest.mark.gpu
def test_two_streams_one_decoder_handle(oracle, dtl, cuda):
    """Two decodes of ONE dtl_ldpc handle in flight on different HIP streams (legal: the decode entry points take a
    const handle and the caller's stream): every launch has its own work-queue counter, no codeword is skipped."""
    import torch
    code = oracle.Ldpc(ALIST_300)
    dec = dtl.ldpc_dec(ALIST_300)
    data_a, llr_a = _noisy_llrs(code, 60000, 0.0, seed=1)
    data_b, llr_b = _noisy_llrs(code, 60000, 0.5, seed=2)
    ref_a = code.decode_batch(llr_a, n_threads=8)
    ref_b = code.decode_batch(llr_b, n_threads=8)
    ta, tb = torch.from_numpy(llr_a).to(cuda), torch.from_numpy(llr_b).to(cuda)
    outs = []
    for rep in range(3):
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        ba = torch.full((60000, code.k), 7, dtype=torch.uint8, device=cuda)
        bb = torch.full((60000, code.k), 7, dtype=torch.uint8, device=cuda)
        na = torch.full((60000,), -1, dtype=torch.int32, device=cuda)
        nb = torch.full((60000,), -1, dtype=torch.int32, device=cuda)
        torch.cuda.synchronize()
        with torch.cuda.stream(sa):
            dec.decode_batch(ta, ba, na)
        with torch.cuda.stream(sb):
            dec.decode_batch(tb, bb, nb)
        torch.cuda.synchronize()
        outs.append((ba.cpu().numpy(), na.cpu().numpy(), bb.cpu().numpy(), nb.cpu().numpy()))
    for ba, na, bb, nb in outs:
        np.testing.assert_array_equal(na, ref_a[1])
        np.testing.assert_array_equal(nb, ref_b[1])
        np.testing.assert_array_equal(ba, ref_a[0])
        np.testing.assert_array_equal(bb, ref_b[0])
