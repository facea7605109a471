"""K10 parity: HIP LDPC sum-product decoder vs the CPU oracle (bit-exact bits and nit)."""
import numpy as np
import pytest

from conftest import ALIST_100_23, ALIST_100_27, ALIST_300


def _noisy_llrs(code, ncw, esn0_db, seed):
    rng = np.random.default_rng(seed)
    data = rng.integers(0, 2, (ncw, code.k)).astype(np.uint8)
    cws = np.stack([code.encode(d) for d in data])
    sig2 = 10 ** (-esn0_db / 10) / 2
    y = (2.0 * cws - 1) + rng.normal(0, np.sqrt(sig2), cws.shape)
    return data, (2 * y / sig2).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("alist,esn0", [(ALIST_300, 0.0), (ALIST_300, 1.5), (ALIST_300, 4.0),
                                        (ALIST_100_23, -3.0), (ALIST_100_27, -2.0)])
def test_decode_batch_matches_oracle(oracle, dtl, cuda, alist, esn0):
    import torch
    code = oracle.Ldpc(alist)
    dec = dtl.ldpc_dec(alist)
    assert (dec.get_n(), dec.get_k()) == (code.n, code.k)
    data, llr = _noisy_llrs(code, 3000, esn0, seed=int(esn0 * 10) + code.n)
    ref_bits, ref_nit = oracle.Ldpc.decode_batch(code, llr, n_threads=8)
    bits, nit = dec.decode_batch(torch.from_numpy(llr).to(cuda))
    torch.cuda.synchronize()
    bits, nit = bits.cpu().numpy(), nit.cpu().numpy()
    assert 0 < ref_nit.mean() < 15, "test point must exercise BP iterations"
    np.testing.assert_array_equal(nit, ref_nit)
    np.testing.assert_array_equal(bits, ref_bits)


@pytest.mark.gpu
def test_noise_free_and_saturated_inputs(oracle, dtl, cuda):
    """Noise-free (nit=0 early exit, the only path the reference's own QA exercises,
    qa_ofdm_adaptive_fec.py:140) and +-inf / huge LLRs (sigma=0.001 demapper default)."""
    import torch
    code = oracle.Ldpc(ALIST_300)
    dec = dtl.ldpc_dec(ALIST_300)
    rng = np.random.default_rng(5)
    data = rng.integers(0, 2, (64, code.k)).astype(np.uint8)
    cws = np.stack([code.encode(d) for d in data]).astype(np.float32)
    for scale in (1.0, 40.0, 800.0, np.inf):
        llr = np.where(cws > 0, scale, -scale).astype(np.float32)
        # flip a few bits so that BP must run on saturated messages
        llr2 = llr.copy()
        llr2[::2, 5] *= -1
        for x in (llr, llr2):
            ref_bits, ref_nit = code.decode_batch(x)
            bits, nit = dec.decode_batch(torch.from_numpy(x).to(cuda))
            np.testing.assert_array_equal(nit.cpu().numpy(), ref_nit)
            np.testing.assert_array_equal(bits.cpu().numpy(), ref_bits)
    bits, nit = dec.decode_batch(torch.from_numpy(np.where(cws > 0, 5.0, -5.0).astype(np.float32)).to(cuda))
    assert (nit.cpu().numpy() == 0).all()
    np.testing.assert_array_equal(bits.cpu().numpy(), data)


@pytest.mark.gpu
def test_gather_descriptor_shortened(oracle, dtl, cuda):
    """tb_decoder::decode layout (lib/dtl/tb_decoder.cc:160-190): [n-k check | k_i data] on air,
    -15 fill for shortened positions, codeword split across two stream segments."""
    import ctypes as C
    import torch
    from gr_dtl_amd._lib import CwDesc
    code = oracle.Ldpc(ALIST_300)
    dec = dtl.ldpc_dec(ALIST_300)
    rng = np.random.default_rng(11)
    ncw, ncheck = 500, code.n - code.k
    k_used = rng.integers(100, code.k + 1, ncw)
    stream, descs, full = [], (CwDesc * ncw)(), np.full((ncw, code.n), -15.0, np.float32)
    pos, out_off = 7, 0
    for i in range(ncw):
        d = np.zeros(code.k, np.uint8)
        d[:k_used[i]] = rng.integers(0, 2, k_used[i])
        cw = code.encode(d)[:ncheck + k_used[i]]
        llr = ((2.0 * cw - 1) * 4 + rng.normal(0, 2.8, cw.size)).astype(np.float32)
        full[i, :cw.size] = llr
        len0 = int(rng.integers(1, cw.size + 1))
        gap = int(rng.integers(0, 50))
        descs[i] = CwDesc(pos, pos + len0 + gap, out_off, len0, cw.size - len0, int(k_used[i]), 0)
        stream.append((pos, llr[:len0]))
        stream.append((pos + len0 + gap, llr[len0:]))
        pos += cw.size + gap + 3
        out_off += int(k_used[i])
    buf = np.full(pos + 16, np.nan, np.float32)
    for p, seg in stream:
        buf[p:p + seg.size] = seg
    ref_bits, ref_nit = code.decode_batch(full)
    d_desc = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(cuda)
    out_bits = torch.zeros(out_off, dtype=torch.uint8, device=cuda)
    out_nit = torch.zeros(ncw, dtype=torch.int32, device=cuda)
    dec.decode_gather(torch.from_numpy(buf).to(cuda), d_desc, out_bits, out_nit)
    got = out_bits.cpu().numpy()
    np.testing.assert_array_equal(out_nit.cpu().numpy(), ref_nit)
    o = 0
    for i in range(ncw):
        np.testing.assert_array_equal(got[o:o + k_used[i]], ref_bits[i, :k_used[i]])
        o += int(k_used[i])
    assert ref_nit.max() > 0
