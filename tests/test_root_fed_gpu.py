"""Root-fed operation (gr_dtl_amd/sharding.py): the stream of one capture is cut into per-rank blocks with a guard
frame on both sides, every block is decoded by its own receiver, and each rank keeps the transport blocks that
start inside its own frames.  The concatenation over the ranks must be what one receiver delivers for the whole
stream.  (The ranks are played by one process here; the exchange itself is covered on CPU with gloo,
tests/test_sharding_cpu.py.)"""
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)


@pytest.mark.gpu
@pytest.mark.parametrize("workload,n_frames,world", [("c2", 700, 2), ("c2", 700, 3), ("c3", 240, 4)])
def test_blocks_of_a_root_fed_stream_tile_the_single_receiver_result(dtl, cuda, workload, n_frames, world):
    import bench
    from gr_dtl_amd.sharding import owned_payload, shard_plan
    # the bench's stream shapes at an Es/N0 where most blocks pass their CRC (there must be payload to compare)
    w = dict(bench.WORKLOADS[workload], esn0_db={"c2": 13.0, "c3": 23.0}[workload])
    cfg = bench.build_config(dtl, w)
    x, frames, n_user, user = bench.synth_stream(dtl, cfg, w, n_frames, cuda, seed=3)
    rx = dtl.ofdm_receiver(cfg)
    whole = rx.work(x)
    ok = whole.tbs[whole.tbs["crc_ok"] == 1]
    assert len(ok) > 0.8 * len(whole.tbs) > 50
    want = whole.payload.cpu().numpy()
    pieces, n_blocks = [], 0
    for rank in range(world):
        plan = shard_plan(frames, rx.frame_samples, 100, x.numel(), world, rank)
        assert plan["stop"] - plan["start"] < x.numel()                  # really a part of the stream
        res = dtl.ofdm_receiver(cfg).work(x[plan["start"]:plan["stop"]])
        data, tbs = owned_payload(res, plan)
        pieces.append(data)
        n_blocks += int((tbs["crc_ok"] == 1).sum())
    got = np.concatenate(pieces)
    assert n_blocks == len(ok)
    np.testing.assert_array_equal(got, want)
    # and it is the transmitted data: every delivered block is a slice of the user bytes
    assert got.size % (ok["user_bits"][0] // 8) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("workload,n_frames,world,n_pieces", [("c2", 900, 2, 6), ("c2", 900, 3, 4), ("c3", 320, 4, 5)])
def test_pipelined_hand_over_equals_one_shot_equals_single_receiver(dtl, cuda, workload, n_frames, world, n_pieces):
    """The root-fed path with the scatter overlapped (sharding.scatter_pieces + decode_pieces: a rank decodes piece k with
    ofdm_receiver.feed while pieces k+1.. arrive) against the one-shot form (the whole block, one work() call) and
    against ONE receiver over the whole stream: the same transport-block records, the same bytes.  The ranks are played
    by one process; the pieces are cut exactly as the exchange cuts them (dtl_shard_pieces)."""
    import bench
    from gr_dtl_amd.sharding import ShardFeed, decode_pieces, owned_payload, shard_pieces, shard_plan
    w = dict(bench.WORKLOADS[workload], esn0_db={"c2": 13.0, "c3": 23.0}[workload])
    cfg = bench.build_config(dtl, w)
    x, frames, n_user, user = bench.synth_stream(dtl, cfg, w, n_frames, cuda, seed=11)
    rx = dtl.ofdm_receiver(cfg)
    whole = rx.work(x)
    want = whole.payload.cpu().numpy()
    ok = whole.tbs[whole.tbs["crc_ok"] == 1]
    assert len(ok) > 0.8 * len(whole.tbs) > 50
    got_one, got_piped, recs_one, recs_piped = [], [], [], []
    for rank in range(world):
        plan = shard_plan(frames, rx.frame_samples, 100, x.numel(), world, rank)
        block = x[plan["start"]:plan["stop"]]
        one = dtl.ofdm_receiver(cfg).work(block)
        bounds = shard_pieces(plan, n_pieces)
        assert len(bounds) - 1 == n_pieces
        piped = decode_pieces(dtl.ofdm_receiver(cfg), ShardFeed(plan, bounds, block, None, []))
        # the streamed calls report what the one call reports (frames, blocks, bytes)
        np.testing.assert_array_equal(piped.frames["trigger"], one.frames["trigger"])
        for k in ("tb_no", "crc_ok", "avg_it", "user_bits", "out_offset"):
            np.testing.assert_array_equal(piped.tbs[k], one.tbs[k], err_msg=k)
        np.testing.assert_array_equal(piped.payload.cpu().numpy(), one.payload.cpu().numpy())
        for res, data_list, rec_list in ((one, got_one, recs_one), (piped, got_piped, recs_piped)):
            data, tbs = owned_payload(res, plan)
            data_list.append(data)
            rec_list.append(tbs)
    np.testing.assert_array_equal(np.concatenate(got_one), want)
    np.testing.assert_array_equal(np.concatenate(got_piped), want)
    assert sum(int((t["crc_ok"] == 1).sum()) for t in recs_piped) == len(ok)
    np.testing.assert_array_equal(np.concatenate([t["tb_no"] for t in recs_piped]), np.concatenate([t["tb_no"] for t in recs_one]))
