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
