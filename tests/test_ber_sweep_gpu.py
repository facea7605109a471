"""BASELINE.json configs[4]: mixed-MCS stream, 1024-point FFT, LDPC, AWGN sweep.  At every swept Es/N0 the HIP chain
and the CPU oracle decode the same noisy stream; frames, transport blocks, payload bytes and therefore the BER /
block-error figures must be identical (the BER curve matches by construction, not within a tolerance)."""
import os
import sys

import numpy as np
import pytest

from helpers import oracle_feedback

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.gpu
@pytest.mark.parametrize("snr_db,gap,arith", [(18.0, 0, "gr_exact"), (24.0, 0, "gr_exact"), (30.0, 0, "gr_exact"),
                                              (22.0, 64, "gr_exact"), (28.0, 64, "gr_exact"), (21.0, 0, "fast"),
                                              (24.0, 0, "fast"),
                                              # the high end of the curve, where the deep sweep (tools/ber_sweep.py --deep,
                                              # profiles/r03/ber_deep_*.json) measures BER 1e-3 .. 1e-5
                                              (26.0, 0, "gr_exact"), (27.5, 0, "gr_exact"), (26.0, 0, "fast"),
                                              (26.0, 64, "fast")])
def test_mixed_mcs_1024_sweep_point_matches_oracle(oracle, dtl, cuda, snr_db, gap, arith):
    import torch
    import ber_sweep
    from conftest import CODES
    from gr_dtl_amd.link_sim import bit_errors
    cfg = ber_sweep.make_config(dtl, 1024, 10, 0.95, sync_correct=(gap == 0), arith=arith)
    acc, y, sent = ber_sweep.sweep_point(dtl, cfg, snr_db, 6, 77, cuda, gap=gap)
    rx = dtl.ofdm_receiver(cfg)
    res = rx.work(torch.from_numpy(y).to(cuda))
    ocfg = oracle.Config(fft_len=cfg.fft_len, cp_len=cfg.cp_len, frame_length=cfg.frame_length, has_fec=True,
                         sync_threshold=cfg.sync_threshold, use_sync_correct=cfg.use_sync_correct,
                         occupied_carriers=cfg.occupied_carriers, pilot_carriers=cfg.pilot_carriers,
                         pilot_symbols=cfg.pilot_symbols, sync_word1=cfg.sync_word1, sync_word2=cfg.sync_word2,
                         arith={"gr_exact": 0, "fast": 1}[arith])
    code = oracle.Ldpc(os.path.join(CODES, "n_0300_k_0152_gap_03.alist"))
    ref = oracle.rx_stream(ocfg, code, y, feedback=oracle_feedback(oracle, cfg, rx), want=("frames",))
    assert len(res.frames) == ref["n_frames"]
    assert [int(t["crc_ok"]) for t in res.tbs] == [int(t["crc_ok"]) for t in ref["tbs"]]
    assert [int(t["avg_it"]) for t in res.tbs] == [int(t["avg_it"]) for t in ref["tbs"]]
    np.testing.assert_array_equal(res.payload.cpu().numpy(), ref["user_bytes"])

    class RefResult:            # the oracle's output in the shape bit_errors() reads
        tbs = res.tbs
        def payload_bytes(self):
            return np.asarray(ref["user_bytes"], dtype=np.uint8)
    ref_acc = bit_errors(sent, RefResult(), 256)
    assert (ref_acc["bit_errors"], ref_acc["blocks_ok"]) == (acc["bit_errors"], acc["blocks_ok"])
    assert acc["blocks_sent"] >= 20
    if snr_db >= 28.0:
        assert acc["bler"] < 0.6    # the high-SNR end of the curve actually decodes


@pytest.mark.gpu
def test_no_fec_frame_dumps_and_ber_report(oracle, dtl, cuda, tmp_path):
    """The no-FEC chain with the reference's frame dumps on both sides: TX dump (user bytes per frame) and RX dump
    (every packed frame) in the frame_file_store format, compared by ber_report (= tools/ber.py's figures)."""
    import torch
    from gr_dtl_amd.link_sim import add_awgn
    from gr_dtl_amd.ofdm_transmitter import ofdm_transmitter
    cfg = dtl.ofdm_adaptive_rx_config(frame_length=20)
    cfg.sync_threshold = 0.95
    tx = ofdm_transmitter(cfg)
    rng = np.random.default_rng(11)
    per = tx.frame_capacity * 2 // 8 - 4
    user = rng.integers(0, 256, per * 60, dtype=np.uint8)
    iq, hdrs = tx.work(user, 2)
    x = np.concatenate([np.zeros(100, np.complex64), iq.reshape(-1), np.zeros(tx.frame_samples, np.complex64)])
    y = add_awgn(x, 64 * 0.25 / 10 ** (11.0 / 10), rng)       # QPSK Es/N0 11 dB: a few frames fail their CRC
    rx = dtl.ofdm_receiver(cfg)
    res = rx.work(torch.from_numpy(y).to(cuda))
    txf, rxf = str(tmp_path / "tx.bin"), str(tmp_path / "rx.bin")
    with dtl.frame_file_store(txf) as st:
        dtl.store_tx_frames(st, user, per)
    with dtl.frame_file_store(rxf) as st:
        dtl.store_rx_frames(st, res)
    rep = dtl.ber_report(txf, rxf, tool_compat=False)   # payload against payload (the tool itself reads 12 bytes late)
    assert rep["frames_sent"] >= 55 and rep["frames_received"] >= 40
    assert rep["mismatch_length"] == 0 and rep["missing_tx"] == 0
    # frames without bit errors are exactly the frames whose CRC-32 passed
    stored = {n for n, _ in dtl.read_frames(rxf)}
    assert rep["crc_ok"] == sum(1 for t in res.tbs if t["crc_ok"] and int(t["tb_no"]) in stored)
    assert 0 <= rep["ber_detected"] < 0.1 and rep["ber_overall"] >= rep["ber_detected"] * 0.99
    # monitor counters agree with the records
    probe = dtl.monitor_probe()
    dec = probe.dec_msgs(res)
    assert dec[-1]["crc_ok_count"] == int((res.tbs["crc_ok"] == 1).sum())
    eq = probe.eq_msgs(res)
    assert len(eq) == res.n_payload_frames and eq[-1]["lost_frames_rate"] >= 0
