"""Frame-batch sharding across the GPUs of one node.

Frames are independent given their own preamble and header, so the multi-GPU path is
pure data parallelism: one process per GPU, contiguous blocks of frames per rank, no data-path
collective.  Only the timing / counters are combined (all_reduce MAX / SUM over RCCL, or gloo on CPU).
"""


def shard_bounds(n_frames, world_size, rank):
    """Contiguous block [lo, hi) of frames for `rank` (frame f -> rank floor(f*G/F))."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    lo = (n_frames * rank) // world_size
    hi = (n_frames * (rank + 1)) // world_size
    return lo, hi


def shard_sample_range(n_frames, frame_samples, lead, world_size, rank, guard_frames=1):
    """Sample range of the rank's shard in a back-to-back frame stream: its frames plus `guard_frames`
    of trailing context so that the Schmidl-Cox window / demux see the whole last frame."""
    lo, hi = shard_bounds(n_frames, world_size, rank)
    start = lead + lo * frame_samples if rank else 0
    stop = lead + min(hi + guard_frames, n_frames) * frame_samples
    return start, stop


def combine(elapsed_s, counters, dist=None, device=None):
    """max-over-ranks time and summed counters; dist is torch.distributed (or None for one process)."""
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    c = torch.tensor([float(v) for v in counters], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(v) for v in c.tolist()]
