"""Phase B (frame equaliser -> serializer -> soft demapper) has three forms that run the same arithmetic:
  general   frame_eq_kernel<false> + demap_kernel   (the equaliser's rows are asked for; what tests/test_rx_gpu.py compares
                                                     with the oracle stage by stage)
  split     frame_eq_kernel<true>  + demap_kernel   (production: the staged equaliser, symbols through HBM)
  fused     frame_eq_kernel<true, false, MODE>      (DTL_RX_FUSED_PHASE_B=1: the demapper in the equaliser's epilogue, the
                                                     frame's symbols parked in LDS / registers; 4.5 GB less HBM traffic per
                                                     C3 step, 3 % slower -- DESIGN.md 10)
Soft bits, serializer output, SNR estimates, block records and bytes must be bit-identical across the three, with integer
carrier offsets (rows shifted by one symbol), every constellation, short and ragged frames, both distance modes and both
arithmetics (lib/dtl/ofdm_adaptive_constellation_soft_cf_impl.cc:89-148, ofdm_adaptive_equalizer.cc:217-268)."""
import os

import numpy as np
import pytest

from helpers import ES, make_pair, make_stream

CASES = [
    # fft_len, frame_length, constellation(s), Es/N0, user bytes, carrier offset (subcarriers), llr_dist_mode, arith
    (64, 20, (2,), 14.0, 9000, 0.0, 0, "gr_exact"),
    (64, 20, (4,), 24.0, 14000, 2.0, 0, "gr_exact"),       # positive integer offset: lanes with lag +1
    (64, 20, (3,), 21.0, 9000, -2.0, 1, "gr_exact"),       # negative: lag -1; plain-distance metric
    (256, 20, (4,), 23.0, 60000, 0.0, 0, "gr_exact"),      # the bench's shape
    (256, 20, (4,), 23.0, 60000, 4.0, 0, "fast"),
    (256, 12, (1, 2, 3, 4), 25.0, 30000, -2.0, 0, "gr_exact"),
    (128, 9, (4, 2), 24.0, 12000, 2.0, 1, "fast"),
    (64, 5, (4,), 24.0, 3000, 0.0, 0, "gr_exact"),         # fewer symbols than the parked rows
    (64, 1, (2,), 14.0, 300, 0.0, 0, "gr_exact"),
    (128, 17, (3,), 22.0, 9000, 2.0, 0, "gr_exact"),
]


def _stream(oracle, ocfg, code, cnsts, n_bytes, esn0, eps, seed):
    rng = np.random.default_rng(seed)
    parts, frame_no = [], 0
    for cn in cnsts:
        user = rng.integers(0, 256, n_bytes // len(cnsts)).astype(np.uint8)
        syms, hdrs = oracle.fec_frames_build(code, 1, cn, ocfg.frame_capacity, user, first_frame_no=frame_no)
        frame_no += len(hdrs)
        parts.append(oracle.tx_frames(ocfg, syms, hdrs).reshape(-1))
    x = np.concatenate([np.zeros(100, np.complex64)] + parts + [np.zeros(ocfg.frame_samples, np.complex64)])
    if eps:
        x = (x * np.exp(2j * np.pi * eps * np.arange(x.size) / ocfg.fft_len)).astype(np.complex64)
    nv = ocfg.fft_len * min(ES[c] for c in cnsts) / 10 ** (esn0 / 10)
    x = x + (rng.normal(0, np.sqrt(nv / 2), x.size) + 1j * rng.normal(0, np.sqrt(nv / 2), x.size))
    return x.astype(np.complex64)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[str(i) for i in range(len(CASES))])
def test_three_forms_of_phase_b_agree_bit_for_bit(oracle, dtl, cuda, case):
    import torch
    fft_len, L, cnsts, esn0, n_bytes, eps, dist_mode, arith = case
    pcfg, ocfg, code = make_pair(oracle, dtl, fft_len=fft_len, frame_length=L, llr_dist_mode=dist_mode, arith=arith)
    x = _stream(oracle, ocfg, code, cnsts, n_bytes, esn0, eps, seed=fft_len + L)
    xt = torch.from_numpy(x).to(cuda)
    want = ("llr", "soft_syms")
    os.environ.pop("DTL_RX_FUSED_PHASE_B", None)
    split = dtl.ofdm_receiver(pcfg).work(xt, want=want)
    general = dtl.ofdm_receiver(pcfg).work(xt, want=want + ("eq_hard", "eq_soft"))
    os.environ["DTL_RX_FUSED_PHASE_B"] = "1"
    try:
        fused = dtl.ofdm_receiver(pcfg).work(xt, want=want)
        fused_quiet = dtl.ofdm_receiver(pcfg).work(xt, want=("llr",))      # the symbols are not written at all
    finally:
        os.environ.pop("DTL_RX_FUSED_PHASE_B", None)
    assert fused.n_payload_frames >= 3 and len(fused.tbs) >= 1
    if eps:
        assert (fused.frames["carr_offset"][fused.frames["header_ok"] == 1] != 0).any()
    for other, name in ((split, "split"), (general, "general"), (fused_quiet, "fused, no symbol output")):
        assert np.array_equal(fused.frames, other.frames), name
        assert np.array_equal(fused.tbs, other.tbs), name
        assert torch.equal(fused.payload, other.payload), name
        assert torch.equal(fused.llr.view(torch.int32), other.llr.view(torch.int32)), name
        if other.soft_syms is not None:
            assert torch.equal(torch.view_as_real(fused.soft_syms).view(torch.int32),
                               torch.view_as_real(other.soft_syms).view(torch.int32)), name
    # and the soft bits are the oracle's (1e-5 relative: north_star; in practice the same bits)
    ref = oracle.rx_stream(ocfg, code, x, want=("frames", "llr"))
    llr = fused.llr.cpu().numpy()
    checked = 0
    for f, rf, rl in zip(fused.frames, ref["frames"], ref["llr"]):
        if not rf["header_ok"] or not rf["hdr"]["payload_len"]:
            continue
        nb = min(rf["n_payload_syms"], ocfg.frame_capacity) * rf["hdr"]["cnst"]
        got, want_l = llr[f["llr_offset"]:f["llr_offset"] + nb], rl[:nb]
        with np.errstate(invalid="ignore"):
            ok = (np.abs(got - want_l) <= 1e-5 * np.abs(want_l)) | (got == want_l)       # (equal infinities)
        assert ok.all(), int(f["frame_no"])
        checked += nb
    assert checked > 0
