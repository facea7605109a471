"""N>1 path on CPU: world_size-2 gloo processes shard a frame batch and combine timing / counters the
way bench.py does over RCCL."""
import os
import socket
import subprocess
import sys

from conftest import ROOT

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["DTL_ROOT"])
import torch, torch.distributed as dist
from gr_dtl_amd.sharding import shard_bounds, shard_sample_range, combine
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n_frames, fs, lead = 1001, 1920, 100
lo, hi = shard_bounds(n_frames, world, rank)
s0, s1 = shard_sample_range(n_frames, fs, lead, world, rank)
# every rank "processes" its shard: count frames, pretend rank r needs (r+1) seconds
elapsed, (frames, samples) = combine(float(rank + 1), [hi - lo, s1 - s0], dist)
owners = torch.zeros(n_frames, dtype=torch.int32)
owners[lo:hi] += 1
dist.all_reduce(owners)
assert int(owners.min()) == 1 and int(owners.max()) == 1, "shards must tile the batch exactly"
assert frames == n_frames and elapsed == float(world)
assert samples >= lead + n_frames * fs
if rank == 0:
    print("OK", frames, elapsed)
dist.destroy_process_group()
'''


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, DTL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "OK 1001.0 2.0" in out.stdout


def test_shard_bounds_properties():
    sys.path.insert(0, ROOT)
    from gr_dtl_amd.sharding import shard_bounds
    for n in (0, 1, 7, 4096, 65536, 524288):
        for w in (1, 2, 4, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


ROOT_FED_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["DTL_ROOT"])
import numpy as np, torch, torch.distributed as dist
from gr_dtl_amd.sharding import shard_plan, scatter_stream, gather_results
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n_frames, fs, lead = 37, 640, 100
total = lead + n_frames * fs + fs
x = None
if rank == 0:
    g = torch.Generator().manual_seed(5)
    x = torch.view_as_complex(torch.randn((total, 2), generator=g))
mine, plan = scatter_stream(x, n_frames, fs, lead, total, dist)
# every rank checks its block against the stream it can regenerate from the seed
g = torch.Generator().manual_seed(5)
full = torch.view_as_complex(torch.randn((total, 2), generator=g))
assert mine.dtype == torch.complex64 and torch.equal(mine, full[plan["start"]:plan["stop"]])
assert plan == shard_plan(n_frames, fs, lead, total, world, rank)
# results back to the root: ragged byte strings + structured records, rank order
dt = np.dtype([("tb_no", "<i4"), ("crc_ok", "<i4"), ("out_offset", "<i8")])
pay = np.arange(100 * (rank + 1) + 3 * rank, dtype=np.int64).astype(np.uint8)
rec = np.zeros(2 + rank, dtype=dt); rec["tb_no"] = np.arange(2 + rank) + 10 * rank; rec["crc_ok"] = 1
gp, gr = gather_results(pay, rec, dist)
if rank == 0:
    want = np.concatenate([np.arange(100 * (r + 1) + 3 * r, dtype=np.int64).astype(np.uint8) for r in range(world)])
    assert np.array_equal(gp, want)
    assert gr.dtype == dt and list(gr["tb_no"]) == [0, 1, 10, 11, 12]
    print("ROOTFED OK", gp.size, len(gr))
else:
    assert gp is None and gr is None
# an empty contribution must not hang the exchange
gp, gr = gather_results(np.zeros(0, dtype=np.uint8) if rank else pay, rec[:0] if rank else rec, dist)
if rank == 0:
    assert gp.size == pay.size and len(gr) == len(rec)
dist.destroy_process_group()
'''


def test_two_rank_gloo_scatter_gather(tmp_path):
    """The exchange of a root-fed receiver: overlapping blocks out (point-to-point), ragged results back."""
    script = tmp_path / "worker_rootfed.py"
    script.write_text(ROOT_FED_WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, DTL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "ROOTFED OK 303 5" in out.stdout


def test_shard_plan_and_block_ownership():
    import numpy as np
    sys.path.insert(0, ROOT)
    from gr_dtl_amd.sharding import owned_blocks, shard_plan
    n_frames, fs, lead = 10, 1000, 100
    total = lead + n_frames * fs + fs
    plans = [shard_plan(n_frames, fs, lead, total, 3, r) for r in range(3)]
    assert plans[0]["start"] == 0 and plans[-1]["stop"] == total
    for a, b in zip(plans, plans[1:]):
        assert a["own_hi"] == b["own_lo"]                       # ownership tiles the stream
        assert a["stop"] == min(total, a["own_hi"] + 3 * fs)  # guard frames behind
        assert b["start"] == max(0, b["own_lo"] - 4 * fs)       # detector warm-up in front
    # rank 1 of 3 owns frames 3..5; say it decoded frames 2..6; every frame starts a block that ends in the next one
    fdt = np.dtype([("trigger", "<i8"), ("payload_index", "<i4"), ("payload_len", "<i4"), ("tb_no", "<i4")])
    tdt = np.dtype([("tb_no", "<i4"), ("crc_ok", "<i4")])
    p = plans[1]
    frames = np.zeros(5, dtype=fdt)
    for i, f in enumerate(range(2, 7)):
        frames[i] = (lead + f * fs + 40 - p["start"], i, 100, f)
    tbs = np.zeros(4, dtype=tdt)                                # blocks 2..5 complete (6 would need frame 7)
    tbs["tb_no"] = [2, 3, 4, 5]
    keep = owned_blocks(frames, tbs, p)
    assert list(tbs["tb_no"][keep]) == [3, 4, 5]                # block 2 started in the left guard frame
    # a header that failed (no payload) and a block that never completed do not shift the matching
    frames2 = frames.copy()
    frames2[2]["payload_index"] = -1                            # frame 4 lost: block 4 never starts, block 3 incomplete
    tbs2 = np.zeros(2, dtype=tdt)
    tbs2["tb_no"] = [2, 5]
    assert list(tbs2["tb_no"][owned_blocks(frames2, tbs2, p)]) == [5]


def test_block_ownership_against_ground_truth_with_losses():
    """Random streams of frames and blocks with lost frames, incomplete blocks, blocks spanning several frames and
    wrapping 8-bit block numbers: the records-only matching of owned_blocks must reproduce the ownership known
    from the construction."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from gr_dtl_amd.sharding import owned_blocks
    fdt = np.dtype([("trigger", "<i8"), ("payload_index", "<i4"), ("payload_len", "<i4"), ("tb_no", "<i4")])
    tdt = np.dtype([("tb_no", "<i4"), ("crc_ok", "<i4")])
    rng = np.random.default_rng(11)
    fs = 1000
    for trial in range(40):
        n_frames = int(rng.integers(50, 900))
        span = float(rng.choice([1.02, 1.4, 0.6, 2.3]))           # block length in frame payloads
        lost = rng.random(n_frames) < rng.choice([0.0, 0.03, 0.15])
        start_sample = int(rng.integers(0, 5000))
        own_lo = start_sample + int(rng.integers(0, n_frames // 2)) * fs
        own_hi = own_lo + int(rng.integers(1, n_frames)) * fs
        plan = dict(start=start_sample, own_lo=own_lo, own_hi=own_hi)
        # block b covers stream positions [b*span, (b+1)*span) in units of frames; it starts in frame floor(b*span)
        n_blocks = int(n_frames / span)
        frames, tb_rows, truth = [], [], []
        first_block_of_frame = {}
        for b in range(n_blocks):
            first_block_of_frame.setdefault(int(np.floor(b * span)), b)
        last_named = 1
        for f in range(n_frames):
            # the header names the (first) block that starts in this frame; a frame inside a long block repeats it
            named = first_block_of_frame.get(f)
            if named is None:
                named = last_named
            last_named = named
            if lost[f]:
                continue
            frames.append((f * fs + 70, len(frames), 100, (named + 1) % 256))
        for b in range(n_blocks):
            f0, f1 = int(np.floor(b * span)), int(np.floor((b + 1) * span - 1e-9))
            if f1 >= n_frames or lost[f0]:
                continue                                           # never started (or runs past the stream)
            # completes if its start frame is there; a lost later frame makes it fail its CRC but it is still recorded
            ok = not lost[f0:f1 + 1].any()
            if first_block_of_frame.get(f0) != b:
                continue                                           # second block starting in one frame: not modelled
            tb_rows.append(((b + 1) % 256, int(ok)))
            truth.append(own_lo <= start_sample + f0 * fs + 70 < own_hi)
        if not tb_rows or not frames:
            continue
        fr = np.array(frames, dtype=fdt)
        tb = np.array(tb_rows, dtype=tdt)
        keep = owned_blocks(fr, tb, plan)
        want = np.nonzero(np.array(truth))[0]
        assert np.array_equal(keep, want), (trial, span, keep[:10], want[:10])
