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


PIECES_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["DTL_ROOT"])
import numpy as np, torch, torch.distributed as dist
from gr_dtl_amd.sharding import (ShardResult, decode_pieces, gather_results, owned_payload, scatter_pieces, scatter_stream,
                                 shard_pieces, shard_plan)
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n_frames, fs, lead = 61, 7680, 100
total = lead + n_frames * fs + fs
g = torch.Generator().manual_seed(9)
full = torch.view_as_complex(torch.randn((total, 2), generator=g))
x = full.clone() if rank == 0 else None
# one-shot and pipelined hand-over deliver the same samples; the pieces tile the rank's block, cut at tile multiples
whole, plan = scatter_stream(x, n_frames, fs, lead, total, dist)
for n_pieces in (1, 3, 8):
    feed = scatter_pieces(x, n_frames, fs, lead, total, dist, n_pieces=n_pieces)
    assert feed.plan == plan == shard_plan(n_frames, fs, lead, total, world, rank)
    assert feed.bounds == shard_pieces(plan, n_pieces) and len(feed) == n_pieces
    assert feed.bounds[0] == plan["start"] and feed.bounds[-1] == plan["stop"]
    assert all((b - plan["start"]) % 4096 == 0 for b in feed.bounds[1:-1])
    got = list(feed)
    feed.finish()
    assert all(p.dtype == torch.complex64 for p in got)
    assert torch.equal(torch.cat(got), whole) and torch.equal(whole, full[plan["start"]:plan["stop"]])

# decode_pieces with a stand-in receiver whose "frames" are the positions of a marker value in the stream: the merged
# records of the streamed calls must be what one call over the whole block reports
class FakeResult:
    def __init__(self, frames, tbs, payload):
        self.frames, self.tbs, self.payload, self.n_payload_frames = frames, tbs, payload, len(frames)
FDT = np.dtype([("trigger", "<i8"), ("payload_index", "<i4"), ("payload_len", "<i4"), ("tb_no", "<i4")])
TDT = np.dtype([("tb_no", "<i4"), ("crc_ok", "<i4"), ("user_bits", "<i4"), ("out_offset", "<i8")])
class FakeRx:
    """a frame starts every `fs` samples behind `lead` of the ORIGINAL stream; block number = frame index mod 256"""
    def __init__(self, start): self.start, self.pos = start, 0
    def reset(self): self.pos = 0
    def feed(self, piece, final=False):
        lo, hi = self.pos, self.pos + piece.numel()
        self.pos = hi
        first = -(-(self.start + lo - lead) // fs)                 # frames whose first sample lies in [lo, hi)
        idx = [f for f in range(max(first, 0), n_frames) if self.start + lo <= lead + f * fs < self.start + hi]
        fr = np.zeros(len(idx), FDT)
        tb = np.zeros(len(idx), TDT)
        for i, f in enumerate(idx):
            fr[i] = (lead + f * fs - self.start, i, 10, f % 256)
            tb[i] = (f % 256, 1, 24, 3 * i)
        pay = torch.tensor([b for f in idx for b in (f % 251, (f * 7) % 251, 255)], dtype=torch.uint8)
        return FakeResult(fr, tb, pay)
feed = scatter_pieces(x, n_frames, fs, lead, total, dist, n_pieces=5)
res = decode_pieces(FakeRx(plan["start"]), feed)
feed.finish()
one = FakeRx(plan["start"]); one.reset()
ref = one.feed(whole, final=True)
assert np.array_equal(res.frames["trigger"], ref.frames["trigger"]) and np.array_equal(res.tbs["tb_no"], ref.tbs["tb_no"])
assert np.array_equal(res.tbs["out_offset"], ref.tbs["out_offset"]) and torch.equal(res.payload, ref.payload)
data, mine = owned_payload(res, plan)
gp, gr = gather_results(data, mine, dist)
if rank == 0:
    want = np.array([b for f in range(n_frames) for b in (f % 251, (f * 7) % 251, 255)], dtype=np.uint8)
    assert np.array_equal(gp, want), (gp.size, want.size)
    assert list(gr["tb_no"]) == [f % 256 for f in range(n_frames)]
    print("PIECES OK", world, gp.size)
dist.destroy_process_group()
'''


def _launch(tmp_path, text, world, name):
    script = tmp_path / name
    script.write_text(text)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, DTL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                          env=env, capture_output=True, text=True, timeout=600)


def test_pipelined_scatter_two_and_three_ranks(tmp_path, dtl):
    """scatter_pieces (the hand-over that overlaps with the decode) == scatter_stream, sample for sample, with 1 / 3 / 8
    pieces; the records of a block decoded piece by piece merge into those of one call; ownership + gather reassemble the
    stream's blocks exactly once -- gloo, world size 2 and 3."""
    for world in (2, 3):
        out = _launch(tmp_path, PIECES_WORKER, world, "worker_pieces.py")
        assert out.returncode == 0, out.stderr[-3000:]
        assert "PIECES OK %d 183" % world in out.stdout


def test_c_abi_sharding_equals_the_numpy_restatement(dtl):
    """dtl_shard_plan / dtl_owned_blocks / dtl_owned_payload_ranges (what a C++ host calls) against an independent numpy
    statement of the same rules, on random shapes."""
    import numpy as np
    from gr_dtl_amd.sharding import owned_blocks, owned_payload, shard_pieces, shard_plan

    def np_plan(n_frames, fs, lead, total, world, rank, guard=3):
        lo, hi = (n_frames * rank) // world, (n_frames * (rank + 1)) // world
        own_lo = lead + lo * fs if rank else 0
        own_hi = lead + hi * fs if rank + 1 < world else total
        start = max(0, own_lo - max(guard, 4) * fs) if rank else 0
        stop = min(total, own_hi + guard * fs) if rank + 1 < world else total
        return dict(start=start, stop=stop, own_lo=min(own_lo, total), own_hi=min(own_hi, total))

    def np_unwrap(v):
        v = np.asarray(v, dtype=np.int64)
        return v if v.size == 0 else np.concatenate([[0], np.cumsum(np.diff(v) % 256)])

    def np_owned(frames, tbs, plan):
        m = (frames["payload_index"] >= 0) & (frames["payload_len"] > 0)
        f_no, f_pos = frames["tb_no"][m].astype(np.int64), frames["trigger"][m].astype(np.int64) + plan["start"]
        if f_no.size == 0 or len(tbs) == 0:
            return np.zeros(0, np.int64)
        first = np.concatenate([[True], f_no[1:] != f_no[:-1]])
        s_no, s_pos = f_no[first], f_pos[first]
        s_seq = np_unwrap(s_no)
        t_no = tbs["tb_no"].astype(np.int64)
        t_seq = np_unwrap(t_no) + (t_no[0] - s_no[0]) % 256
        at = np.searchsorted(s_seq, t_seq)
        hit = (at < s_seq.size) & (s_seq[np.minimum(at, s_seq.size - 1)] == t_seq)
        pos = np.where(hit, s_pos[np.minimum(at, s_seq.size - 1)], -1)
        idx = np.maximum.accumulate(np.where(hit, np.arange(len(tbs)), -1))
        pos = np.where(idx >= 0, pos[np.maximum(idx, 0)], -1)
        return np.nonzero((pos >= plan["own_lo"]) & (pos < plan["own_hi"]))[0].astype(np.int64)

    rng = np.random.default_rng(3)
    for _ in range(200):
        n_frames, fs, lead = int(rng.integers(0, 5000)), int(rng.choice([1920, 7680, 17920])), int(rng.integers(0, 300))
        total = lead + n_frames * fs + int(rng.integers(0, 2 * fs))
        world = int(rng.integers(1, 9))
        guard = int(rng.choice([1, 3, 5]))
        for rank in range(world):
            p = shard_plan(n_frames, fs, lead, total, world, rank, guard)
            assert p == np_plan(n_frames, fs, lead, total, world, rank, guard)
            b = shard_pieces(p, int(rng.integers(1, 12)))
            assert b[0] == p["start"] and b[-1] == p["stop"] and all(x < y for x, y in zip(b, b[1:])) or p["stop"] == p["start"]
    fdt = np.dtype([("trigger", "<i8"), ("payload_index", "<i4"), ("payload_len", "<i4"), ("tb_no", "<i4")])
    tdt = np.dtype([("tb_no", "<i4"), ("crc_ok", "<i4"), ("user_bits", "<i4"), ("out_offset", "<i8")])
    for _ in range(300):
        nf = int(rng.integers(0, 400))
        fr = np.zeros(nf, fdt)
        fr["trigger"] = np.sort(rng.integers(0, 10 ** 6, nf))
        fr["payload_index"] = np.where(rng.random(nf) < 0.1, -1, 1)
        fr["payload_len"] = np.where(rng.random(nf) < 0.05, 0, 100)
        fr["tb_no"] = (int(rng.integers(0, 256)) + np.cumsum(rng.integers(0, 3, nf))) % 256
        nt = int(rng.integers(0, 300))
        tb = np.zeros(nt, tdt)
        tb["tb_no"] = (int(rng.integers(0, 256)) + np.cumsum(rng.integers(0, 3, nt))) % 256
        tb["crc_ok"] = rng.random(nt) < 0.8
        tb["user_bits"] = 8 * rng.integers(1, 40, nt)
        off = np.concatenate([[0], np.cumsum(np.where(tb["crc_ok"] == 1, tb["user_bits"] // 8, 0))[:-1]]) if nt else np.zeros(0)
        tb["out_offset"] = np.where(tb["crc_ok"] == 1, off, -1)
        lo = int(rng.integers(0, 10 ** 6))
        plan = dict(start=int(rng.integers(0, 1000)), stop=10 ** 7, own_lo=lo, own_hi=lo + int(rng.integers(1, 10 ** 6)))
        keep = owned_blocks(fr, tb, plan)
        assert np.array_equal(keep, np_owned(fr, tb, plan))

        class R:
            frames, tbs, payload = fr, tb, np.arange(int(np.where(tb["crc_ok"] == 1, tb["user_bits"] // 8, 0).sum()) if nt else 0, dtype=np.int64).astype(np.uint8)
        data, mine = owned_payload(R, plan)
        ok = mine[mine["crc_ok"] == 1]
        want = np.concatenate([R.payload[int(o):int(o + n // 8)] for o, n in zip(ok["out_offset"], ok["user_bits"])] or [R.payload[:0]])
        assert np.array_equal(data, want)
