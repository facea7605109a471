"""Batched transmitter on the GPU (dtl_tx_* / ofdm_transmitter.work_device) against the CPU oracle's TX:
framing (symbol indices, header fields) and the on-air samples must be identical -- the IFFT runs the same
radix-2 butterflies on the same twiddle table, so the IQ stream is compared bit for bit -- and a noise-free
loop TX(device) -> RX(device) returns the user bytes."""
import numpy as np
import pytest

from helpers import make_pair

HDR_KEYS = ("payload_len", "frame_no", "cnst", "fb_cnst", "tb_no", "fec_fb", "tb_offset", "fec", "tb_payload")


def _same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("fft_len,frame_length,cnst,nbytes", [(64, 20, 1, 1500), (64, 20, 2, 4000), (64, 20, 3, 5000),
                                                              (64, 20, 4, 9000), (64, 6, 2, 1200), (256, 10, 4, 20000),
                                                              (1024, 4, 3, 15000), (64, 20, 2, 1)])
def test_fec_tx_matches_oracle(oracle, dtl, cuda, fft_len, frame_length, cnst, nbytes):
    import torch
    from gr_dtl_amd.ofdm_transmitter import ofdm_transmitter
    pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=fft_len, frame_length=frame_length)
    rng = np.random.default_rng(500 + fft_len + cnst)
    user = rng.integers(0, 256, nbytes, dtype=np.uint8)
    ref_syms, ref_hdrs = oracle.fec_frames_build(code, 1, cnst, ocfg.frame_capacity, user, first_frame_no=4090)
    ref_iq = oracle.tx_frames(ocfg, ref_syms, ref_hdrs)
    tx = ofdm_transmitter(pcfg)
    iq, hdrs, syms = tx.work_device(torch.from_numpy(user).to(cuda), cnst, "fec_1", first_frame_no=4090, want_syms=True)
    assert len(hdrs) == len(ref_hdrs)
    for k in HDR_KEYS:
        np.testing.assert_array_equal(hdrs[k], [h[k] for h in ref_hdrs], err_msg=k)
    np.testing.assert_array_equal(syms.cpu().numpy(), ref_syms)
    got = iq.cpu().numpy()
    assert got.shape == ref_iq.shape
    assert _same_bits(got, ref_iq)
    # the host (numpy) transmitter of the package frames the same way
    h_iq, h_hdrs = tx.work(user, cnst, "fec_1", first_frame_no=4090)
    assert [h["payload_len"] for h in h_hdrs] == list(hdrs["payload_len"])
    np.testing.assert_allclose(h_iq, got, rtol=0, atol=2e-5 * fft_len ** 0.5)


@pytest.mark.gpu
@pytest.mark.parametrize("cnst,nbytes", [(1, 700), (2, 3000), (3, 2000), (4, 6001)])
def test_no_fec_tx_matches_oracle(oracle, dtl, cuda, cnst, nbytes):
    import torch
    from gr_dtl_amd.ofdm_transmitter import ofdm_transmitter
    pcfg, ocfg, code = make_pair(oracle, dtl, fec=None)
    rng = np.random.default_rng(600 + cnst)
    user = rng.integers(0, 256, nbytes, dtype=np.uint8)
    ref_syms, ref_hdrs = oracle.nofec_frames_build(cnst, ocfg.frame_capacity, user, first_frame_no=7)
    ref_iq = oracle.tx_frames(ocfg, ref_syms, ref_hdrs)
    tx = ofdm_transmitter(pcfg)
    iq, hdrs, syms = tx.work_device(torch.from_numpy(user).to(cuda), cnst, first_frame_no=7, want_syms=True)
    for k in ("payload_len", "frame_no", "cnst", "fb_cnst"):
        np.testing.assert_array_equal(hdrs[k], [h[k] for h in ref_hdrs], err_msg=k)
    np.testing.assert_array_equal(syms.cpu().numpy(), ref_syms)
    assert _same_bits(iq.cpu().numpy(), ref_iq)


@pytest.mark.gpu
def test_device_loopback_and_gap(oracle, dtl, cuda):
    """TX(device) -> zeros in front / between / behind -> RX(device): every user byte comes back; numbering
    continues across calls (mixed-MCS stream)."""
    import torch
    from gr_dtl_amd.ofdm_transmitter import ofdm_transmitter
    pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=64, frame_length=20, use_sync_correct=False)
    tx = ofdm_transmitter(pcfg)
    rx = dtl.ofdm_receiver(pcfg)
    g = torch.Generator(device=cuda)
    g.manual_seed(5)
    parts, users, frame_no, tb_no = [torch.zeros(100, dtype=torch.complex64, device=cuda)], [], 0, 0
    for cnst, nbytes in ((2, 3000), (4, 5000), (1, 800), (3, 2500)):
        user = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device=cuda, generator=g)
        iq, hdrs, _ = tx.work_device(user, cnst, "fec_1", first_frame_no=frame_no, first_tb_no=tb_no, gap=48)
        assert iq.shape[1] == tx.frame_samples + 48 and torch.all(iq[:, tx.frame_samples:] == 0)
        frame_no = (frame_no + len(hdrs)) & 0xFFF
        tb_no = int(hdrs["tb_no"][-1])
        parts.append(iq.reshape(-1))
        users.append(user)
    parts.append(torch.zeros(tx.frame_samples, dtype=torch.complex64, device=cuda))
    res = rx.work(torch.cat(parts))
    assert int((res.tbs["crc_ok"] == 1).sum()) == len(res.tbs)
    assert torch.equal(res.payload, torch.cat(users))
    assert list(res.frames["frame_no"][res.frames["payload_index"] >= 0]) == list(range(frame_no))


@pytest.mark.gpu
def test_tx_errors(dtl, cuda):
    import torch
    from gr_dtl_amd.ofdm_transmitter import ofdm_transmitter
    tx = ofdm_transmitter(dtl.ofdm_adaptive_rx_config())
    with pytest.raises(dtl._lib.DtlError):
        tx.work_device(torch.zeros(10, dtype=torch.uint8), 2)              # host tensor: no CPU fallback
    with pytest.raises(dtl._lib.DtlError, match="Unknown constellation"):
        tx.work_device(torch.zeros(10, dtype=torch.uint8, device=cuda), 9)
    iq, hdrs, _ = tx.work_device(torch.zeros(0, dtype=torch.uint8, device=cuda), 2)  # nothing to send
    assert iq.shape[0] == 0 and len(hdrs) == 0
