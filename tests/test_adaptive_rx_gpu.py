"""dtl.ofdm_adaptive_rx -- the reference's outer API (python/dtl/ofdm_adaptive_rx.py:32-59, grc/dtl_ofdm_adaptive_rx.block.yml):
built with from_parameters(config_dict=...) from an examples/config_fec.json-style dictionary, one complex input, the
seven output streams of the hier block + the monitor port.  Every port is compared with the oracle's restatement of the
stream the reference graph puts on it."""
import numpy as np
import pytest

from helpers import make_stream, oracle_feedback

# the "ofdm_config" object of the reference's examples/config_fec.json (what the GRC block passes as config_dict)
CONFIG_FEC = {
    "mcs": [[-100000, ["bpsk", "fec_1"]], [11, ["qpsk", "fec_1"]], [16, ["psk8", "fec_1"]], [21, ["qam16", "fec_1"]]],
    "initial_mcs_id": 1,
    "fec_codes": [["fec_1", "n_0300_k_0152_gap_03.alist"]],
}


def _oracle_side(oracle, cfg):
    import os
    from conftest import CODES
    ocfg = oracle.Config(fft_len=cfg.fft_len, cp_len=cfg.cp_len, frame_length=cfg.frame_length, has_fec=True,
                         sync_threshold=cfg.sync_threshold, use_sync_correct=cfg.use_sync_correct,
                         occupied_carriers=cfg.occupied_carriers, pilot_carriers=cfg.pilot_carriers,
                         pilot_symbols=cfg.pilot_symbols, sync_word1=cfg.sync_word1, sync_word2=cfg.sync_word2)
    return ocfg, oracle.Ldpc(os.path.join(CODES, "n_0300_k_0152_gap_03.alist"))


@pytest.mark.gpu
@pytest.mark.parametrize("cnst,esn0", [(2, 13.0), (4, 23.0)])
def test_from_parameters_and_all_seven_ports(oracle, dtl, cuda, cnst, esn0):
    import torch
    rx = dtl.ofdm_adaptive_rx.from_parameters(config_dict=CONFIG_FEC, packet_length_tag_key="len_tag_key")
    cfg = rx.config
    assert cfg.packet_length_tag_key == "len_tag_key" and cfg.initial_mcs_id == 1 and len(cfg.fec_codes) == 1
    assert [int(m[1][0]) for m in cfg.mcs] == [1, 2, 3, 4]
    ocfg, code = _oracle_side(oracle, cfg)
    user, hdrs, x = make_stream(oracle, ocfg, code, cnst, 9000, esn0, seed=70 + cnst)
    ports = rx.work(torch.from_numpy(x).to(cuda))
    assert len(ports) == 7
    ref = oracle.rx_stream(ocfg, code, x, feedback=oracle_feedback(oracle, cfg, rx.direct_rx),
                           want=("frames", "trig", "angle", "eq_hard", "eq_soft"))
    # 0: decoded bytes
    np.testing.assert_array_equal(ports[0].cpu().numpy(), ref["user_bytes"])
    assert sum(t["crc_ok"] for t in ref["tbs"]) >= 10
    # 1: feedback transmitter -- not part of the accelerated path
    assert ports[1].numel() == 0 and ports[1].dtype == torch.complex64
    # 2, 3: trigger byte streams, one item per input sample
    assert ports[2].dtype == torch.uint8 and ports[2].numel() == x.size
    np.testing.assert_array_equal(ports[2].cpu().numpy(), ref["trig_fixed"])
    np.testing.assert_array_equal(ports[3].cpu().numpy(), ref["trig_raw"])
    # 4, 5: payload equaliser outputs of the frames whose header was accepted, in order
    ok = [i for i, f in enumerate(ref["frames"]) if f["header_ok"]]
    L, N = cfg.frame_length, cfg.fft_len
    want_hard = np.concatenate([ref["eq_hard"][i].reshape(L, N) for i in ok])
    want_soft = np.concatenate([ref["eq_soft"][i].reshape(-1) for i in ok])
    assert tuple(ports[4].shape) == (len(ok) * L, N)
    np.testing.assert_array_equal(ports[4].cpu().numpy(), want_hard)
    got_soft = ports[5].cpu().numpy()
    assert got_soft.shape == want_soft.shape
    assert (got_soft.view(np.uint32) == want_soft.view(np.uint32)).all()
    # 6: fine frequency offset, sampled at the raw triggers and held
    got_f, want_f = ports[6].cpu().numpy(), ref["angle"]
    assert got_f.shape == want_f.shape
    np.testing.assert_array_equal(got_f.view(np.uint32), np.asarray(want_f, np.float32).view(np.uint32))   # identical bits (tests/test_rx_gpu.py)
    assert len(np.unique(want_f)) > 5
    # monitor port: one equaliser message per decoded frame, one decoder message per transport block
    eq_msgs = [m for m in ports.monitor if "estimated_snr_tag_key" in m]
    dec_msgs = [m for m in ports.monitor if "tb_no" in m]
    assert len(eq_msgs) == len(ok) and len(dec_msgs) == len(ref["tbs"])
    assert dec_msgs[-1]["crc_ok_count"] == sum(t["crc_ok"] for t in ref["tbs"])
    assert [m["constellation_key"] for m in eq_msgs] == [ref["frames"][i]["fb_cnst"] for i in ok]


@pytest.mark.gpu
def test_streamed_buffers_give_the_same_ports(oracle, dtl, cuda):
    """feed(): the hier block over consecutive scheduler buffers -- concatenated port streams equal one work() call."""
    import torch
    rx = dtl.ofdm_adaptive_rx.from_parameters(config_dict=CONFIG_FEC, use_sync_correct=False)
    ocfg, code = _oracle_side(oracle, rx.config)
    user, hdrs, x = make_stream(oracle, ocfg, code, 2, 7000, 13.0, seed=5)
    whole = rx.work(torch.from_numpy(x).to(cuda))
    rx2 = dtl.ofdm_adaptive_rx.from_parameters(config_dict=CONFIG_FEC, use_sync_correct=False)
    cuts = [0, 3001, 3002, 9000, x.size // 2 + 13, x.size]
    parts = [rx2.feed(torch.from_numpy(x[a:b]).to(cuda), final=(b == x.size)) for a, b in zip(cuts, cuts[1:])]
    for port in (0, 2, 3, 4, 5, 6):
        got = torch.cat([p[port] for p in parts])
        assert got.shape == whole[port].shape, port
        assert torch.equal(got, whole[port]), port
    assert len([m for p in parts for m in p.monitor]) == len(whole.monitor)
