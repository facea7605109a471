"""Frame-batch sharding across the GPUs of one node.

Frames are independent given their own preamble and header, so the multi-GPU path is
pure data parallelism: one process per GPU, contiguous blocks of frames per rank, no collective inside
the decode.  A receiver whose ranks are fed directly (one capture stream per GPU, bench.py's case) only
combines timing / counters (all_reduce MAX / SUM over RCCL, or gloo on CPU).  When ONE rank holds the
stream, the only exchange the path has is at its two ends: `scatter_stream` hands every rank its block of
samples (point-to-point sends from the root: with RCCL, one xGMI link per peer, all of them at once) and
`gather_results` brings the decoded bytes and the transport-block records back.  Blocks overlap (a warm-up of a few
frames in front, guard frames behind); `owned_blocks` keeps, on every rank, the transport blocks that START inside its own
frames, so that the concatenation over the ranks is what one receiver would have delivered.
"""
import numpy as np


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


# ------------------------------------------------------------------------------------------ root-fed operation

WARMUP_FRAMES = 4  # frame_detect_bb is in sync after three consistent triggers (kSyncedTh, ..._frame_detect_bb_impl.cc:21)


def shard_plan(n_frames, frame_samples, lead, total_samples, world_size, rank, guard_frames=3):
    """-> dict(start, stop, own_lo, own_hi): the samples rank `rank` decodes ([start, stop): its frames, WARMUP_FRAMES
    in front so that the trigger-fixing state machine is in sync when the rank's own frames begin -- only then does
    it re-insert a missed preamble the way the single receiver does -- and `guard_frames` behind: a transport block
    that starts in the rank's last frame must be complete.  A block slightly larger than a frame's payload touches
    three frames (tail of one, a whole one, head of the next), and frames are cut at their trigger, about a symbol
    behind the nominal frame start -- hence three frames; blocks of several frame payloads need more) and the samples it OWNS ([own_lo,
    own_hi): a transport block belongs to the rank whose range holds the trigger of the frame it starts in)."""
    lo, hi = shard_bounds(n_frames, world_size, rank)
    own_lo = lead + lo * frame_samples if rank else 0
    own_hi = lead + hi * frame_samples if rank + 1 < world_size else total_samples
    start = max(0, own_lo - max(guard_frames, WARMUP_FRAMES) * frame_samples) if rank else 0
    stop = min(total_samples, own_hi + guard_frames * frame_samples) if rank + 1 < world_size else total_samples
    return dict(start=int(start), stop=int(stop), own_lo=int(own_lo), own_hi=int(own_hi))


def scatter_stream(x_root, n_frames, frame_samples, lead, total_samples, dist, device=None, src=0, guard_frames=3):
    """Root holds the complex64 stream `x_root` (1-D tensor; None on the other ranks); every rank returns its block
    (shard_plan) as a complex64 tensor on `device` together with the plan.  Point-to-point: the root posts one
    isend per peer (the blocks differ in length and overlap, so this is not a plain scatter), the peers one irecv;
    with the nccl backend these are RCCL send/recv pairs over the direct xGMI links, all in flight together."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    plan = shard_plan(n_frames, frame_samples, lead, total_samples, world, rank, guard_frames)
    if world == 1:
        return x_root[plan["start"]:plan["stop"]], plan
    if rank == src:
        flat = torch.view_as_real(x_root.contiguous())          # (n, 2) float32 view, no copy
        ops = []
        for r in range(world):
            if r == src:
                continue
            p = shard_plan(n_frames, frame_samples, lead, total_samples, world, r, guard_frames)
            ops.append(dist.P2POp(dist.isend, flat[p["start"]:p["stop"]], r))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        mine = x_root[plan["start"]:plan["stop"]]
        for q in reqs:
            q.wait()
        return mine, plan
    buf = torch.empty((plan["stop"] - plan["start"], 2), dtype=torch.float32, device=device)
    for q in dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf, src)]):
        q.wait()
    return torch.view_as_complex(buf), plan


def _unwrap(numbers, modulus=256):
    """Block numbers as a non-decreasing sequence (the header field wraps; consecutive records are < one wrap apart)."""
    numbers = np.asarray(numbers, dtype=np.int64)
    if numbers.size == 0:
        return numbers
    return np.concatenate([[0], np.cumsum(np.diff(numbers) % modulus)])


def owned_blocks(frames, tbs, plan):
    """Indices into `tbs` of the transport blocks this rank owns.  `frames` / `tbs` are the record arrays of a
    work() call over the rank's block (trigger positions relative to plan['start']).  A frame's header names the
    block that starts in it (`tb_no`; `tb_offset` is the tail of the previous block) and blocks complete in stream
    order, so both lists are turned into running block counts (the 8-bit number unwrapped) and matched by value.
    Starts whose block never completed simply find no record; a record without a start of its own (a block flushed
    twice around a lost frame) goes where the record before it went."""
    if len(tbs) == 0:
        return np.zeros(0, dtype=np.int64)
    m = (frames["payload_index"] >= 0) & (frames["payload_len"] > 0)
    f_no = frames["tb_no"][m].astype(np.int64)
    f_pos = frames["trigger"][m].astype(np.int64) + plan["start"]
    if f_no.size == 0:
        return np.zeros(0, dtype=np.int64)
    first = np.concatenate([[True], f_no[1:] != f_no[:-1]])          # a frame that continues a block starts none
    s_no, s_pos = f_no[first], f_pos[first]
    s_seq = _unwrap(s_no)
    t_no = tbs["tb_no"].astype(np.int64)
    t_seq = _unwrap(t_no) + (t_no[0] - s_no[0]) % 256                # the first record's block started <= 255 starts in
    at = np.searchsorted(s_seq, t_seq)
    hit = (at < s_seq.size) & (s_seq[np.minimum(at, s_seq.size - 1)] == t_seq)
    pos = np.where(hit, s_pos[np.minimum(at, s_seq.size - 1)], -1)
    # records without a start inherit the previous record's position
    idx = np.where(hit, np.arange(len(tbs)), -1)
    idx = np.maximum.accumulate(idx)
    pos = np.where(idx >= 0, pos[np.maximum(idx, 0)], -1)
    return np.nonzero((pos >= plan["own_lo"]) & (pos < plan["own_hi"]))[0].astype(np.int64)


def owned_payload(result, plan, on_device=False):
    """-> (bytes, tbs records) of the CRC-ok transport blocks this rank owns, in stream order.  The bytes are a uint8
    numpy array, or with on_device=True a slice of the result's device tensor (no copy through the host)."""
    tbs = result.tbs
    keep = owned_blocks(result.frames, tbs, plan)
    mine = tbs[keep]
    ok = mine[mine["crc_ok"] == 1]
    payload = result.payload
    if not on_device:
        payload = payload.cpu().numpy() if hasattr(payload, "cpu") else np.asarray(payload)
    if len(ok) == 0:
        return payload[:0], mine
    offs, lens = ok["out_offset"].astype(np.int64), ok["user_bits"].astype(np.int64) // 8
    if np.all(offs[1:] == offs[:-1] + lens[:-1]):                      # the usual case: one contiguous run of blocks
        return payload[int(offs[0]):int(offs[-1] + lens[-1])], mine
    parts = [payload[int(o):int(o + n)] for o, n in zip(offs, lens)]
    if on_device:
        import torch
        return torch.cat(parts), mine
    return np.concatenate(parts), mine


def gather_results(payload, records, dist, device=None, dst=0):
    """Decoded bytes and transport-block records of every rank on `dst`, in rank order (= stream order).
    payload: 1-D uint8 numpy array or torch tensor (a device tensor stays on the device: RCCL moves it GPU to GPU);
    records: numpy structured array.  Lengths differ per rank: they are all-gathered first (16 bytes per rank), then
    every peer sends exactly its bytes.  -> (payload, records) on dst -- the payload of the kind that was passed
    in -- and (None, None) elsewhere."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    if world == 1:
        return payload, records
    as_tensor = hasattr(payload, "is_cuda")
    pay_t = payload if as_tensor else torch.from_numpy(np.ascontiguousarray(payload))
    rec_t = torch.from_numpy(np.ascontiguousarray(records).view(np.uint8).reshape(-1).copy())
    if device is not None:
        pay_t, rec_t = pay_t.to(device), rec_t.to(device)
    pay_t = pay_t.contiguous()
    sizes = torch.tensor([pay_t.numel(), rec_t.numel()], dtype=torch.int64, device=device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    if rank != dst:
        ops = [dist.P2POp(dist.isend, t, dst) for t in (pay_t, rec_t) if t.numel()]
        for q in (dist.batch_isend_irecv(ops) if ops else []):
            q.wait()
        return None, None
    pay_bufs, rec_bufs, ops = {dst: pay_t}, {dst: rec_t}, []
    for r in range(world):
        if r == dst:
            continue
        n_pay, n_rec = int(all_sizes[r][0].item()), int(all_sizes[r][1].item())
        pay_bufs[r] = torch.empty(n_pay, dtype=torch.uint8, device=device)
        rec_bufs[r] = torch.empty(n_rec, dtype=torch.uint8, device=device)
        if n_pay:
            ops.append(dist.P2POp(dist.irecv, pay_bufs[r], r))
        if n_rec:
            ops.append(dist.P2POp(dist.irecv, rec_bufs[r], r))
    for q in (dist.batch_isend_irecv(ops) if ops else []):
        q.wait()
    pay = torch.cat([pay_bufs[r] for r in range(world)])
    rec = np.concatenate([rec_bufs[r].cpu().numpy().view(records.dtype) for r in range(world)])
    return (pay if as_tensor else pay.cpu().numpy()), rec
