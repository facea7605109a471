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


@pytest.mark.gpu
@pytest.mark.parametrize("arith", ["gr_exact", "fast"])
@pytest.mark.parametrize("alist", [ALIST_300, ALIST_100_23, ALIST_100_27])
def test_gather_descriptors_partial_reception(oracle, dtl, cuda, alist, arith):
    """Every shape a codeword's descriptor can take (lib/dtl/tb_decoder.cc:143-195): whole, cut in two anywhere with a
    gap in the stream, received only in part (the rest of the sent positions count as 0), not received at all,
    shortened down to one data bit; clean codewords (no iteration: the decoder's short path) mixed with noisy ones.
    The decoder reads soft bits where they lie in the stream, so the stream is NaN outside the received pieces."""
    import torch
    from gr_dtl_amd._lib import CwDesc
    code = oracle.Ldpc(alist)
    dec = dtl.ldpc_dec(alist, arith=arith)
    rng = np.random.default_rng(code.n * 7 + code.k + len(arith))
    ncw, ncheck = 1500, code.n - code.k
    descs, full = (CwDesc * ncw)(), np.full((ncw, code.n), -15.0, np.float32)
    pieces, pos, out_off, k_useds = [], 5, 0, []
    for i in range(ncw):
        k_used = int(rng.integers(1, code.k + 1)) if i % 3 else code.k
        d = np.zeros(code.k, np.uint8)
        d[:k_used] = rng.integers(0, 2, k_used)
        want = ncheck + k_used
        cw = code.encode(d)[:want]
        noise = (0.0, 1.5, 2.8)[i % 3 if i % 7 else 2]
        llr = ((2.0 * cw - 1) * 4 + rng.normal(0, 1, want) * noise).astype(np.float32)
        how = i % 5
        recv = want if how < 3 else (int(rng.integers(1, want)) if how == 3 else (0 if i % 10 == 4 else int(rng.integers(1, want))))
        full[i, :recv] = llr[:recv]
        full[i, recv:want] = 0.0
        len0 = recv if how == 0 else int(rng.integers(0, recv + 1))
        if recv > 0 and len0 == 0 and i % 2:
            len0 = 1
        gap = int(rng.integers(0, 40))
        off0 = pos if len0 > 0 or i % 4 else 0          # (a codeword whose start was never received: offset 0, length 0)
        descs[i] = CwDesc(off0, pos + len0 + gap, out_off, len0, recv - len0, k_used, 0)
        pieces.append((pos, llr[:len0]))
        pieces.append((pos + len0 + gap, llr[len0:recv]))
        pos += recv + gap + int(rng.integers(0, 4))
        out_off += k_used
        k_useds.append(k_used)
    buf = np.full(pos + 8, np.nan, np.float32)
    buf[0] = 3.0                                         # offset 0 is read (and dropped) for pieces of length 0
    for p, seg in pieces:
        buf[p:p + seg.size] = seg
    ref_bits, ref_nit = code.decode_batch(full, n_threads=8, **({"arith": oracle.FAST} if arith == "fast" else {}))
    d_desc = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(cuda)
    out_bits = torch.full((out_off,), 7, dtype=torch.uint8, device=cuda)
    out_nit = torch.full((ncw,), -1, dtype=torch.int32, device=cuda)
    dec.decode_gather(torch.from_numpy(buf).to(cuda), d_desc, out_bits, out_nit)
    got = out_bits.cpu().numpy()
    np.testing.assert_array_equal(out_nit.cpu().numpy(), ref_nit)
    o = 0
    for i in range(ncw):
        np.testing.assert_array_equal(got[o:o + k_useds[i]], ref_bits[i, :k_useds[i]], err_msg=f"codeword {i}")
        o += k_useds[i]
    assert (ref_nit == 0).mean() > 0.1 and (ref_nit > 0).mean() > 0.2


@pytest.mark.gpu
def test_every_batch_size_is_decoded_once(oracle, dtl, cuda):
    """The decoder's work queue (chunks dealt round-robin to eight ticket counters, big chunks first, two-codeword
    chunks at the end, the next ticket requested a chunk ahead): every codeword of a batch is decoded exactly once
    whatever the batch size -- fewer codewords than counters, than waves, than one chunk per wave, one more or less
    than a multiple of the chunk sizes."""
    import torch
    code = oracle.Ldpc(ALIST_300)
    dec = dtl.ldpc_dec(ALIST_300)
    sizes = [1, 2, 3, 7, 8, 9, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1000, 4095, 4096, 4097, 12287, 12289, 32769, 65537, 200001]
    data, llr = _noisy_llrs(code, max(sizes), 5.0, seed=77)   # ~15 % of the codewords need no iteration
    ref_bits, ref_nit = oracle.Ldpc.decode_batch(code, llr[:8192], n_threads=8)
    d_llr = torch.from_numpy(llr).to(cuda)
    full_bits = full_nit = None
    for n in reversed(sizes):
        bits = torch.full((n, code.k), 9, dtype=torch.uint8, device=cuda)
        nit = torch.full((n,), -7, dtype=torch.int32, device=cuda)
        dec.decode_batch(d_llr[:n], bits, nit)
        bits, nit = bits.cpu().numpy(), nit.cpu().numpy()
        if full_bits is None:
            full_bits, full_nit = bits, nit               # the largest batch: checked against the oracle on its first 8192 ...
            np.testing.assert_array_equal(nit[:8192], ref_nit)
            np.testing.assert_array_equal(bits[:8192], ref_bits)
            assert (nit >= 0).all() and (bits <= 1).all()
            assert 0.05 < (nit == 0).mean() < 0.95
        else:                                             # ... and every smaller one against its prefix
            np.testing.assert_array_equal(nit, full_nit[:n], err_msg=f"batch of {n}")
            np.testing.assert_array_equal(bits, full_bits[:n], err_msg=f"batch of {n}")


@pytest.mark.gpu
def test_descriptor_that_received_nothing_reads_no_stale_offset(oracle, dtl, cuda):
    """A codeword with len0 == len1 == 0 (it lies in a gap of the stream) whose offsets are stale -- far outside the caller's
    buffer: the decoder must not dereference them (it reads soft bit 0 of the stream and drops it).  Decoded like the
    oracle decodes "nothing received": every sent position 0, shortened positions -15."""
    import torch
    from gr_dtl_amd._lib import CwDesc
    code = oracle.Ldpc(ALIST_300)
    dec = dtl.ldpc_dec(ALIST_300)
    rng = np.random.default_rng(12)
    n, k = code.n, code.k
    ncw = 70                                              # more than one wave's worth of tickets
    descs, full = (CwDesc * ncw)(), np.zeros((ncw, n), np.float32)
    buf = np.zeros(ncw * n + 16, np.float32)
    pos = 0
    for i in range(ncw):
        cw = code.encode(rng.integers(0, 2, k).astype(np.uint8))
        llr = ((2.0 * cw - 1) * 3 + rng.normal(0, 1, n)).astype(np.float32)
        if i % 9 == 4:                                    # received nothing; offsets point 4 GB - 240 B into nowhere
            descs[i] = CwDesc(0xFFFFFFF0, 0xFFFFFF00, i * k, 0, 0, k, 0)
            full[i] = 0.0
        else:
            buf[pos:pos + n] = llr
            descs[i] = CwDesc(pos, 0, i * k, n, 0, k, 0)
            full[i] = llr
            pos += n
    ref_bits, ref_nit = code.decode_batch(full, n_threads=4)
    d_desc = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(cuda)
    bits = torch.zeros((ncw, k), dtype=torch.uint8, device=cuda)
    nit = torch.zeros(ncw, dtype=torch.int32, device=cuda)
    dec.decode_gather(torch.from_numpy(buf[:pos + 8]).to(cuda), d_desc, bits, nit)
    np.testing.assert_array_equal(nit.cpu().numpy(), ref_nit)
    np.testing.assert_array_equal(bits.cpu().numpy(), ref_bits)
