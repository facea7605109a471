"""Parity at BASELINE.json's full sizes.
C2 (64-FFT, QPSK, LDPC(300,152), 4k-frame batch): compared directly with the oracle.
C3 (256-FFT, 16-QAM, 64k-frame batch, 503 M samples): the oracle would need minutes, so the full-size
run is checked through size-independent properties: every transport block that passes its CRC-32 carries
exactly the bytes that were transmitted in it (encode -> noise -> decode round trip), trigger indices are
strictly increasing and frame-periodic, a second pass over the same buffer is bit-identical (idempotence),
and a 1024-frame prefix of the same stream matches the oracle exactly."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT
from helpers import make_pair, make_stream, oracle_feedback

sys.path.insert(0, ROOT)


@pytest.mark.gpu
def test_c1_plumbing_full_size_matches_oracle(oracle, dtl, cuda):
    """configs[0] on the GPU path as well: 64-point FFT, BPSK, no FEC, 1000 frames, Es/N0 20 dB -- frames, CRC flags and
    bytes identical to the oracle; noise-free: every user byte back."""
    import torch
    pcfg, ocfg, code = make_pair(oracle, dtl, fec=None)
    per = ocfg.frame_capacity // 8 - 4
    for esn0 in (None, 20.0):
        user, hdrs, x = make_stream(oracle, ocfg, None, 1, 1000 * per, esn0, seed=1)
        assert len(hdrs) == 1000
        rx = dtl.ofdm_receiver(pcfg)
        res = rx.work(torch.from_numpy(x).to(cuda))
        ref = oracle.rx_stream(ocfg, None, x, feedback=oracle_feedback(oracle, pcfg, rx), want=("frames",))
        assert len(res.frames) == ref["n_frames"] >= 1000
        assert [(int(t["tb_no"]), int(t["crc_ok"])) for t in res.tbs] == [(t["tb_no"], t["crc_ok"]) for t in ref["tbs"]]
        np.testing.assert_array_equal(res.payload.cpu().numpy(), ref["user_bytes"])
        if esn0 is None:
            np.testing.assert_array_equal(ref["user_bytes"], user)


@pytest.mark.gpu
def test_c2_full_batch_matches_oracle(oracle, dtl, cuda):
    import torch
    pcfg, ocfg, code = make_pair(oracle, dtl)
    user, hdrs, x = make_stream(oracle, ocfg, code, 2, 4096 * 118, 12.0, seed=2)   # ~4k frames
    assert len(hdrs) >= 4000
    rx = dtl.ofdm_receiver(pcfg)
    res = rx.work(torch.from_numpy(x).to(cuda), want=("cw",))
    ref = oracle.rx_stream(ocfg, code, x, feedback=oracle_feedback(oracle, pcfg, rx), want=("frames", "trig"))
    np.testing.assert_array_equal(res.raw_triggers, np.nonzero(ref["trig_raw"])[0])
    np.testing.assert_array_equal(res.fixed_triggers, np.nonzero(ref["trig_fixed"])[0])
    assert len(res.frames) == ref["n_frames"] >= 3500
    for f, rf in zip(res.frames, ref["frames"]):
        assert f["header_ok"] == rf["header_ok"] and f["trigger"] == rf["trigger"]
        if rf["header_ok"]:
            assert (f["frame_no"], f["tb_no"], f["tb_offset"], f["payload_len"]) == \
                (rf["hdr"]["frame_no"], rf["hdr"]["tb_no"], rf["hdr"]["tb_offset"], rf["hdr"]["payload_len"])
            assert (f["feedback_cnst"], f["lost_frames"]) == (rf["fb_cnst"], rf["lost_frames"])
    assert [(t["tb_no"], t["crc_ok"], t["avg_it"]) for t in res.tbs] == [(t["tb_no"], t["crc_ok"], t["avg_it"]) for t in ref["tbs"]]
    np.testing.assert_array_equal(res.payload.cpu().numpy(), ref["user_bytes"])
    assert 0 < res.cw_nit.float().mean().item() < 15


@pytest.mark.gpu
def test_c3_full_size_properties(oracle, dtl, cuda):
    import torch
    import bench
    # the bench's C3 stream at the bench line's own operating point (23 dB: >= 99 % of the frames decoded and >= 90 % of
    # the transport blocks CRC-ok) -- pinned here so that a change of the bench's point does not silently
    # take the round-trip property's material away
    w = dict(bench.WORKLOADS["c3"], esn0_db=23.0)
    assert w["sync_threshold"] == 0.95
    cfg = bench.build_config(dtl, w)
    rx = dtl.ofdm_receiver(cfg)
    x, n_frames, n_user, user_t = bench.synth_stream(dtl, cfg, w, w["frames"], cuda, seed=0)
    assert n_frames >= 65000 and x.numel() > 5.0e8
    res = rx.work(x)
    payload = res.payload.cpu().numpy()
    user = user_t.cpu().numpy()                 # every frame carries its own random payload
    tb_bytes = 52 * 152 // 8 - 4
    # -- triggers: strictly increasing, dominated by the frame period
    d = np.diff(res.raw_triggers)
    assert (d > 0).all() and np.median(d) == rx.frame_samples
    assert res.n_payload_frames >= 0.99 * n_frames
    # -- every CRC-ok TB carries exactly what was sent in it
    ok = res.tbs[res.tbs["crc_ok"] == 1]
    assert len(ok) >= 0.85 * len(res.tbs) > 40000
    # the header carries the block number modulo 256: unwrap it along the stream (fewer than 256 blocks are
    # ever lost in a row), then compare each delivered block with the bytes that were sent in that block
    idx = np.zeros(len(res.tbs), dtype=np.int64)
    last_no, last_idx = 0, 0
    for i, no in enumerate(res.tbs["tb_no"]):
        last_idx += (int(no) - last_no) % 256
        last_no = int(no)
        idx[i] = last_idx
    checked = 0
    sel = np.nonzero(res.tbs["crc_ok"] == 1)[0]
    for i in sel[:: max(1, len(sel) // 20000)]:
        t = res.tbs[i]
        want = user[(idx[i] - 1) * tb_bytes:idx[i] * tb_bytes]
        got = payload[t["out_offset"]:t["out_offset"] + t["user_bits"] // 8]
        assert got.size == want.size and np.array_equal(got, want), (i, t)
        checked += 1
    assert checked > 10000
    assert int(ok["user_bits"].sum()) // 8 == payload.size
    # -- idempotence: a second pass over the same buffer gives the same bytes and records
    res2 = rx.work(x)
    assert torch.equal(res2.payload, res.payload)
    assert np.array_equal(res2.tbs, res.tbs) and np.array_equal(res2.frames["snr_db"], res.frames["snr_db"])
    # -- a 1024-frame prefix against the oracle
    npre = 100 + 1024 * rx.frame_samples + rx.frame_samples // 2
    xp = x[:npre].cpu().numpy()
    resp = rx.work(torch.from_numpy(xp).to(cuda))
    ocfg = oracle.Config(fft_len=cfg.fft_len, cp_len=cfg.cp_len, frame_length=cfg.frame_length, has_fec=True,
                         sync_threshold=cfg.sync_threshold, occupied_carriers=cfg.occupied_carriers,
                         pilot_carriers=cfg.pilot_carriers, pilot_symbols=cfg.pilot_symbols,
                         sync_word1=cfg.sync_word1, sync_word2=cfg.sync_word2)
    code = oracle.Ldpc(os.path.join(ROOT, "gr_dtl_amd", "codes", "n_0300_k_0152_gap_03.alist"))
    ref = oracle.rx_stream(ocfg, code, xp, feedback=oracle_feedback(oracle, cfg, rx))
    assert [(t["tb_no"], t["crc_ok"], t["avg_it"]) for t in resp.tbs] == [(t["tb_no"], t["crc_ok"], t["avg_it"]) for t in ref["tbs"]]
    np.testing.assert_array_equal(resp.payload.cpu().numpy(), ref["user_bytes"])


@pytest.mark.gpu
@pytest.mark.parametrize("arith", ["gr_exact", "fast"])
@pytest.mark.parametrize("esn0", [18.0, 23.0])
def test_c3_bench_operating_points_match_oracle_codeword_by_codeword(oracle, dtl, cuda, arith, esn0):
    """The first 1024 frames of the bench's own C3 stream at the operating points the bench quotes: 23 dB (the headline:
    the link delivers payload) and 18 dB (extra.other_operating_points: the decoder's worst case -- 8.5 iterations per
    codeword, 40 % of the codewords at the 15-iteration limit, many of them on the saturation / NaN path of the
    likelihood-ratio arithmetic).  Frames, block records and the iteration count of EVERY codeword against the oracle,
    in both arithmetics (lib/dtl/ldpc_dec.cc:62-71, tb_decoder.cc:143-195)."""
    import torch
    import bench
    w = dict(bench.WORKLOADS["c3"], esn0_db=esn0)
    cfg = bench.build_config(dtl, w)
    cfg.arith = arith
    rx = dtl.ofdm_receiver(cfg)
    # the same generator calls as the bench (seed = rank 0): the prefix of the 65 536-frame stream depends on the batch
    # size only through the user bytes drawn, which are drawn first -- so build the whole stream and cut
    x, n_frames, n_user, _ = bench.synth_stream(dtl, cfg, w, w["frames"], cuda, seed=0)
    npre = 100 + 1024 * rx.frame_samples + rx.frame_samples // 2
    xp = x[:npre].cpu().numpy()
    del x
    res = rx.work(torch.from_numpy(xp).to(cuda), want=("cw",))
    ocfg = oracle.Config(fft_len=cfg.fft_len, cp_len=cfg.cp_len, frame_length=cfg.frame_length, has_fec=True,
                         sync_threshold=cfg.sync_threshold, occupied_carriers=cfg.occupied_carriers,
                         pilot_carriers=cfg.pilot_carriers, pilot_symbols=cfg.pilot_symbols,
                         sync_word1=cfg.sync_word1, sync_word2=cfg.sync_word2, arith={"gr_exact": 0, "fast": 1}[arith])
    code = oracle.Ldpc(os.path.join(ROOT, "gr_dtl_amd", "codes", "n_0300_k_0152_gap_03.alist"))
    ref = oracle.rx_stream(ocfg, code, xp, feedback=oracle_feedback(oracle, cfg, rx), want=("frames", "trig", "cw"))
    np.testing.assert_array_equal(res.raw_triggers, np.nonzero(ref["trig_raw"])[0])
    np.testing.assert_array_equal(res.fixed_triggers, np.nonzero(ref["trig_fixed"])[0])
    assert len(res.frames) == ref["n_frames"] >= 1015
    for f, rf in zip(res.frames, ref["frames"]):
        assert f["header_ok"] == rf["header_ok"] and f["trigger"] == rf["trigger"]
        if rf["header_ok"]:
            assert (f["frame_no"], f["tb_no"], f["tb_offset"], f["payload_len"]) == \
                (rf["hdr"]["frame_no"], rf["hdr"]["tb_no"], rf["hdr"]["tb_offset"], rf["hdr"]["payload_len"])
            assert (f["feedback_cnst"], f["lost_frames"]) == (rf["fb_cnst"], rf["lost_frames"])
            assert f["snr_db"] == pytest.approx(rf["snr_db"], rel=1e-6)
    assert [(t["tb_no"], t["crc_ok"], t["avg_it"]) for t in res.tbs] == [(t["tb_no"], t["crc_ok"], t["avg_it"]) for t in ref["tbs"]]
    nit = res.cw_nit.cpu().numpy()
    assert nit.size == ref["cw_nit"].size > 50000
    np.testing.assert_array_equal(nit, ref["cw_nit"])
    np.testing.assert_array_equal(res.payload.cpu().numpy(), ref["user_bytes"])
    if esn0 == 18.0:
        # the test must stay at the hard point: a third of the codewords or more run into the iteration limit
        assert (nit == 15).mean() > 0.3 and nit.mean() > 7.0
    else:
        assert (res.tbs["crc_ok"] == 1).mean() > 0.85
