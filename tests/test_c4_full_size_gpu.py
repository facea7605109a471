"""BASELINE.json configs[3] at its full size on ONE GPU: the 524 288-frame, 4.03-Gsample (32 GB) stream of the 8-GPU
root-fed run is built the way bench.root_fed_record builds it (eight 65 536-frame pieces from the GPU transmitter, block
and frame numbers restarting per piece), cut with shard_plan(world=8) exactly as scatter_stream would, and every shard is
decoded by its own receiver; owned_payload keeps what each rank would send back.  The reference for "what one receiver
delivers" is ONE receiver that takes the whole stream through feed() (the streaming form, 16 calls cut at arbitrary
sample positions).  The ranks are played by one process; the exchange itself is covered by tests/test_sharding_cpu.py
(gloo, world size 2) and tools/root_fed.py.  Rests on the reference's per-frame reset of the equaliser state
(lib/dtl/ofdm_adaptive_frame_equalizer_vcvc_impl.cc:117-118,180)."""
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)

WORLD = 8
FRAMES_PER_GPU = 65536


def _unwrap_from_one(tb_no):
    """running block count along one transmitter run (block numbers start at 1 and wrap at 256)"""
    idx = np.zeros(len(tb_no), dtype=np.int64)
    last_no, last_idx = 0, 0
    for i, no in enumerate(tb_no):
        last_idx += (int(no) - last_no) % 256
        last_no = int(no)
        idx[i] = last_idx
    return idx


@pytest.mark.gpu
def test_c4_512k_frames_cut_into_8_shards_tile_the_single_receiver_result(dtl, cuda):
    import torch
    import bench
    from gr_dtl_amd.sharding import owned_payload, shard_plan
    free, total = torch.cuda.mem_get_info()
    if free < 90 * 2 ** 30:
        pytest.skip("needs ~90 GB of free HBM (32 GB stream + one shard's receiver buffers)")
    w = dict(bench.WORKLOADS["c3"])
    assert w["esn0_db"] == 23.0 and w["sync_threshold"] == 0.95
    cfg = bench.build_config(dtl, w)
    rx0 = dtl.ofdm_receiver(cfg)
    fs = rx0.frame_samples
    # ---- the stream of bench.root_fed_record, assembled in place
    users, n_frames, pieces = [], 0, []
    for r in range(WORLD):
        xr, nf, _, user = bench.synth_stream(dtl, cfg, w, FRAMES_PER_GPU, cuda, seed=1000 + r, lead=(r == 0), tail=(r == WORLD - 1))
        pieces.append(xr)
        users.append(user.cpu().numpy())
        n_frames += nf
    total_samples = sum(p.numel() for p in pieces)
    x = torch.empty(total_samples, dtype=torch.complex64, device=cuda)
    at = 0
    for p in pieces:
        x[at:at + p.numel()] = p
        at += p.numel()
    del pieces, xr
    torch.cuda.empty_cache()
    assert n_frames >= 8 * 65500 and total_samples > 4.0e9            # configs[3]: 512k frames, 4.03 Gsamples
    tb_bytes = 52 * 152 // 8 - 4

    # ---- one receiver over the whole stream (streaming form; cuts at arbitrary positions, not on frame boundaries)
    n_calls = 16
    cuts = [int(total_samples * i / n_calls) + (37 * i if 0 < i < n_calls else 0) for i in range(n_calls + 1)]
    one_bytes, one_tbs, one_frames = [], [], 0
    for i in range(n_calls):
        res = rx0.feed(x[cuts[i]:cuts[i + 1]], final=(i == n_calls - 1))
        one_bytes.append(res.payload.cpu().numpy())
        one_tbs.append(res.tbs[["tb_no", "crc_ok", "avg_it", "user_bits"]].copy())
        one_frames += int(res.n_payload_frames)
        del res
    del rx0
    one_bytes = np.concatenate(one_bytes)
    one_tbs = np.concatenate(one_tbs)
    assert one_frames >= 0.99 * n_frames
    assert (one_tbs["crc_ok"] == 1).mean() >= 0.85                    # 23 dB: the link delivers

    # ---- the same stream cut for 8 ranks (aligned with the transmitter restarts) and for 5 (not aligned)
    for world in (WORLD, 5):
        got_bytes, got_tbs = [], []
        for rank in range(world):
            plan = shard_plan(n_frames, fs, 100, total_samples, world, rank)
            assert 0 <= plan["start"] <= plan["own_lo"] < plan["own_hi"] <= plan["stop"] <= total_samples
            assert plan["stop"] - plan["start"] < total_samples / world + 8 * fs
            rx = dtl.ofdm_receiver(cfg)                               # every rank has its own receiver
            res = rx.work(x[plan["start"]:plan["stop"]])
            data, tbs = owned_payload(res, plan)
            got_bytes.append(np.array(data, copy=True))
            got_tbs.append(tbs[["tb_no", "crc_ok", "avg_it", "user_bits"]].copy())
            if world == WORLD:
                # every CRC-ok block this rank owns carries exactly the bytes transmitted in it (piece `rank` of the
                # stream = this rank's own frames; block numbers restart at 1 per piece)
                ok = np.nonzero(tbs["crc_ok"] == 1)[0]
                idx = _unwrap_from_one(tbs["tb_no"])
                offs = np.concatenate([[0], np.cumsum(tbs["user_bits"][ok] // 8)])
                assert offs[-1] == data.size
                user = users[rank]
                step = max(1, len(ok) // 3000)
                for j in range(0, len(ok), step):
                    i = ok[j]
                    want = user[(idx[i] - 1) * tb_bytes:idx[i] * tb_bytes]
                    assert np.array_equal(data[offs[j]:offs[j + 1]], want), (rank, int(i), int(tbs["tb_no"][i]))
            del res, rx
        all_tbs = np.concatenate(got_tbs)
        # blocks tile the single receiver's list without gap or duplicate: the same block numbers in the same order
        assert len(all_tbs) == len(one_tbs), (world, len(all_tbs), len(one_tbs))
        np.testing.assert_array_equal(all_tbs["tb_no"], one_tbs["tb_no"])
        np.testing.assert_array_equal(all_tbs["user_bits"], one_tbs["user_bits"])
        # A shard's oscillator starts at phase 0 where the single receiver's has integrated the held angles of the whole
        # stream before it (frequency_modulator_fc, python/dtl/ofdm_receiver.py:75,89): every sample of a shard is turned
        # by one common angle, which the channel estimate takes out again -- up to float rounding.  Soft bits therefore
        # differ in their last bits, and a codeword on the edge of convergence may need one iteration more or fewer; a
        # block's CRC verdict changes for a handful in half a million.
        diff_ok = all_tbs["crc_ok"] != one_tbs["crc_ok"]
        diff_it = all_tbs["avg_it"] != one_tbs["avg_it"]
        print("world %d: %d blocks, CRC verdict differs in %d, average iteration count in %d"
              % (world, len(all_tbs), int(diff_ok.sum()), int(diff_it.sum())))
        assert diff_ok.mean() < 2e-4 and diff_it.mean() < 2e-3, (world, int(diff_ok.sum()), int(diff_it.sum()))
        # the delivered bytes: block by block wherever both deliver (a CRC-32 vouches for each side)
        both = (all_tbs["crc_ok"] == 1) & (one_tbs["crc_ok"] == 1)
        a_bytes, o_bytes = np.concatenate(got_bytes), one_bytes
        a_off = np.concatenate([[0], np.cumsum(np.where(all_tbs["crc_ok"] == 1, all_tbs["user_bits"] // 8, 0))])
        o_off = np.concatenate([[0], np.cumsum(np.where(one_tbs["crc_ok"] == 1, one_tbs["user_bits"] // 8, 0))])
        assert a_off[-1] == a_bytes.size and o_off[-1] == o_bytes.size
        if not diff_ok.any():
            assert np.array_equal(a_bytes, o_bytes)
        else:
            for i in np.nonzero(both)[0][::7]:
                assert np.array_equal(a_bytes[a_off[i]:a_off[i + 1]], o_bytes[o_off[i]:o_off[i + 1]]), (world, int(i))
