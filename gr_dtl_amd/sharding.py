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
import ctypes as C

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
# Where to cut and which blocks a rank owns is host arithmetic of the C ABI (include/dtl_rx.h: dtl_shard_plan,
# dtl_shard_pieces, dtl_owned_blocks, dtl_owned_payload_ranges -- a C++ host shards with the same calls); the functions
# below are thin callers.  The exchange itself is torch.distributed point-to-point (backend nccl == RCCL over xGMI).

WARMUP_FRAMES = 4  # frame_detect_bb is in sync after three consistent triggers (kSyncedTh, ..._frame_detect_bb_impl.cc:21)
PIECE_ALIGN = 4096  # the sync kernel's tile: pieces of a shard are cut at multiples of it behind the shard's start


class _Shard(C.Structure):   # dtl_shard
    _fields_ = [("start", C.c_int64), ("stop", C.c_int64), ("own_lo", C.c_int64), ("own_hi", C.c_int64)]


def _clib():
    from . import _lib
    L = _lib.lib()
    if not getattr(L, "_shard_declared", False):
        L.dtl_shard_plan.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_Shard)]
        L.dtl_shard_pieces.argtypes = [C.POINTER(_Shard), C.c_int32, C.c_int64, C.c_void_p, C.POINTER(C.c_int32)]
        L.dtl_owned_blocks.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(_Shard), C.c_void_p, C.c_int64,
                                       C.POINTER(C.c_int64)]
        L.dtl_owned_payload_ranges.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                               C.POINTER(C.c_int64)]
        L._shard_declared = True
    return L, _lib


def _as_records(arr, dtype):
    """record array -> contiguous array of the C layout `dtype` (fields copied by NAME; fields the caller's array does not
    have stay zero)"""
    arr = np.asarray(arr)
    if arr.dtype == dtype:
        return np.ascontiguousarray(arr)
    out = np.zeros(len(arr), dtype=dtype)
    for name in (arr.dtype.names or ()):
        if name in dtype.names:
            out[name] = arr[name]
    return out


def _as_shard(plan):
    return _Shard(int(plan["start"]), int(plan.get("stop", plan["own_hi"])), int(plan["own_lo"]), int(plan["own_hi"]))


def shard_plan(n_frames, frame_samples, lead, total_samples, world_size, rank, guard_frames=3):
    """-> dict(start, stop, own_lo, own_hi): the samples rank `rank` decodes ([start, stop): its frames, WARMUP_FRAMES
    in front so that the trigger-fixing state machine is in sync when the rank's own frames begin -- only then does
    it re-insert a missed preamble the way the single receiver does -- and `guard_frames` behind: a transport block
    that starts in the rank's last frame must be complete.  A block slightly larger than a frame's payload touches
    three frames (tail of one, a whole one, head of the next), and frames are cut at their trigger, about a symbol
    behind the nominal frame start -- hence three frames; blocks of several frame payloads need more) and the samples it OWNS ([own_lo,
    own_hi): a transport block belongs to the rank whose range holds the trigger of the frame it starts in).
    (dtl_shard_plan of the C ABI)"""
    L, _lib = _clib()
    out = _Shard()
    _lib.check(L.dtl_shard_plan(int(n_frames), int(frame_samples), int(lead), int(total_samples), int(world_size), int(rank),
                                int(guard_frames), C.byref(out)))
    return dict(start=int(out.start), stop=int(out.stop), own_lo=int(out.own_lo), own_hi=int(out.own_hi))


def shard_pieces(plan, n_pieces, align=PIECE_ALIGN):
    """-> ascending absolute sample positions [start, ..., stop]: the shard cut into at most n_pieces consecutive pieces
    (dtl_shard_pieces: cuts at multiples of `align` behind the shard's start)."""
    L, _lib = _clib()
    bounds = np.zeros(int(n_pieces) + 1, dtype=np.int64)
    n = C.c_int32(0)
    sh = _as_shard(plan)
    _lib.check(L.dtl_shard_pieces(C.byref(sh), int(n_pieces), int(align), bounds.ctypes.data, C.byref(n)))
    return [int(v) for v in bounds[:n.value + 1]]


def scatter_stream(x_root, n_frames, frame_samples, lead, total_samples, dist, device=None, src=0, guard_frames=3):
    """Root holds the complex64 stream `x_root` (1-D tensor; None on the other ranks); every rank returns its block
    (shard_plan) as a complex64 tensor on `device` together with the plan.  Point-to-point: the root posts one
    isend per peer (the blocks differ in length and overlap, so this is not a plain scatter), the peers one irecv;
    with the nccl backend these are RCCL send/recv pairs over the direct xGMI links, all in flight together.
    One-shot form: the whole block arrives before anything is decoded (scatter_pieces overlaps the two)."""
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


class ShardFeed:
    """A rank's block arriving piece by piece (scatter_pieces).  Iterating yields the pieces in stream order, each as
    soon as it has arrived -- while the caller decodes piece k (ofdm_receiver.feed), pieces k+1 .. are in flight: every
    receive was posted up front, each into its own slice of the rank's buffer (memory is not the constraint: a C3 shard
    is 4 GB of 288), and RCCL works through them on its own stream."""

    def __init__(self, plan, bounds, buf, reqs, sends):
        self.plan, self.bounds, self._buf, self._reqs, self._sends = plan, bounds, buf, reqs, sends

    def __len__(self):
        return len(self.bounds) - 1

    def __iter__(self):
        import torch
        start = self.plan["start"]
        for k in range(len(self)):
            if self._reqs is not None:
                for q in self._reqs[k]:
                    q.wait()          # nccl: the current stream waits for piece k; gloo: the host does
            a, b = self.bounds[k] - start, self.bounds[k + 1] - start
            piece = self._buf[a:b]
            yield torch.view_as_complex(piece) if piece.dtype == torch.float32 else piece

    def finish(self):
        """root: the sends have all been handed over (call after the own pieces are decoded)"""
        for q in self._sends:
            q.wait()
        self._sends = []


def scatter_pieces(x_root, n_frames, frame_samples, lead, total_samples, dist, device=None, src=0, guard_frames=3, n_pieces=8):
    """The pipelined form of scatter_stream ("scatter of chunk i+1 overlapped with compute of chunk i"): every
    rank's block is cut into the same number of pieces (shard_pieces); the root posts the sends PIECE-major -- piece 0 to
    every peer, then piece 1, ... -- so that all peers can start decoding after 1/n_pieces of the transfer; a peer posts
    all its receives at once.  -> ShardFeed.  Legal because the receiver's streaming call returns the same frames whatever
    the cuts (dtl_rx_process_stream), which in turn rests on the reference's per-frame reset of the equaliser state
    (lib/dtl/ofdm_adaptive_frame_equalizer_vcvc_impl.cc:117-118,180)."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    plan = shard_plan(n_frames, frame_samples, lead, total_samples, world, rank, guard_frames)
    bounds = shard_pieces(plan, n_pieces)
    if world == 1:
        return ShardFeed(plan, bounds, x_root[plan["start"]:plan["stop"]], None, [])
    if rank == src:
        flat = torch.view_as_real(x_root.contiguous())
        peers = []
        for r in range(world):
            if r == src:
                continue
            p = shard_plan(n_frames, frame_samples, lead, total_samples, world, r, guard_frames)
            peers.append((r, shard_pieces(p, n_pieces)))
        sends = []
        for k in range(max([len(b) - 1 for _, b in peers] + [0])):
            ops = [dist.P2POp(dist.isend, flat[b[k]:b[k + 1]], r) for r, b in peers if k + 1 < len(b)]
            if ops:
                sends.extend(dist.batch_isend_irecv(ops))
        return ShardFeed(plan, bounds, x_root[plan["start"]:plan["stop"]], None, sends)
    buf = torch.empty((plan["stop"] - plan["start"], 2), dtype=torch.float32, device=device)
    reqs = []
    for k in range(len(bounds) - 1):
        a, b = bounds[k] - plan["start"], bounds[k + 1] - plan["start"]
        reqs.append(dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf[a:b], src)]))
    return ShardFeed(plan, bounds, buf, reqs, [])


class ShardResult:
    """The records of a shard decoded piece by piece, merged into what ONE call over the shard returns: frames and block
    records concatenated (trigger positions of feed() are absolute within the shard already; a block's output offset is
    moved behind the bytes of the earlier calls), payload bytes concatenated."""

    def __init__(self):
        self._frames, self._tbs, self._payload = [], [], []
        self.n_payload_bytes = 0
        self.n_payload_frames = 0

    def add(self, res):
        tbs = res.tbs.copy()
        tbs["out_offset"] = np.where(tbs["out_offset"] >= 0, tbs["out_offset"] + self.n_payload_bytes, -1)
        self._frames.append(res.frames.copy())
        self._tbs.append(tbs)
        self._payload.append(res.payload)
        self.n_payload_bytes += int(res.payload.numel() if hasattr(res.payload, "numel") else len(res.payload))
        self.n_payload_frames += int(res.n_payload_frames)

    @property
    def frames(self):
        return np.concatenate(self._frames) if self._frames else np.zeros(0)

    @property
    def tbs(self):
        return np.concatenate(self._tbs) if self._tbs else np.zeros(0)

    @property
    def payload(self):
        import torch
        if self._payload and hasattr(self._payload[0], "numel"):
            return torch.cat(self._payload) if len(self._payload) > 1 else self._payload[0]
        return np.concatenate(self._payload) if self._payload else np.zeros(0, np.uint8)


def decode_pieces(rx, feed, device=None):
    """ofdm_receiver.feed() over the pieces of a ShardFeed as they arrive -> ShardResult (what rx.work(whole block)
    returns, assembled from the streamed calls)."""
    rx.reset()
    acc = ShardResult()
    n = len(feed)
    for k, piece in enumerate(feed):
        if device is not None and not piece.is_cuda:
            piece = piece.to(device)                          # (gloo rehearsal: the exchange is staged through the host)
        acc.add(rx.feed(piece, final=(k + 1 == n)))
    if n == 0:
        rx.reset()
    return acc


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
    twice around a lost frame) goes where the record before it went.  (dtl_owned_blocks of the C ABI)"""
    from .ofdm_receiver import _FRAME_DTYPE, _TB_DTYPE
    if len(tbs) == 0:
        return np.zeros(0, dtype=np.int64)
    L, _lib = _clib()
    fr = _as_records(frames, _FRAME_DTYPE)
    tb = _as_records(tbs, _TB_DTYPE)
    idx = np.zeros(len(tb), dtype=np.int64)
    n = C.c_int64(0)
    sh = _as_shard(plan)
    _lib.check(L.dtl_owned_blocks(fr.ctypes.data if len(fr) else None, len(fr), tb.ctypes.data, len(tb), C.byref(sh),
                                  idx.ctypes.data, len(idx), C.byref(n)))
    return idx[:n.value].copy()


def owned_payload(result, plan, on_device=False):
    """-> (bytes, tbs records) of the CRC-ok transport blocks this rank owns, in stream order.  The bytes are a uint8
    numpy array, or with on_device=True a slice of the result's device tensor (no copy through the host).
    (dtl_owned_blocks + dtl_owned_payload_ranges of the C ABI)"""
    from .ofdm_receiver import _TB_DTYPE
    tbs = result.tbs
    keep = owned_blocks(result.frames, tbs, plan)
    mine = tbs[keep]
    payload = result.payload
    if not on_device:
        payload = payload.cpu().numpy() if hasattr(payload, "cpu") else np.asarray(payload)
    if len(keep) == 0:
        return payload[:0], mine
    L, _lib = _clib()
    tb = _as_records(tbs, _TB_DTYPE)
    offs, lens = np.zeros(len(keep), np.int64), np.zeros(len(keep), np.int64)
    n = C.c_int64(0)
    keep64 = np.ascontiguousarray(keep, dtype=np.int64)
    _lib.check(L.dtl_owned_payload_ranges(tb.ctypes.data, len(tb), keep64.ctypes.data, len(keep64), offs.ctypes.data,
                                          lens.ctypes.data, len(offs), C.byref(n)))
    if n.value == 0:
        return payload[:0], mine
    if n.value == 1:                                                    # the usual case: one contiguous run of blocks
        return payload[int(offs[0]):int(offs[0] + lens[0])], mine
    parts = [payload[int(o):int(o + m)] for o, m in zip(offs[:n.value], lens[:n.value])]
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
