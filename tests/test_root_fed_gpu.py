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
