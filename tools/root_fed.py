#!/usr/bin/env python3
"""Root-fed receiver over the GPUs of one node: rank 0 holds the IQ stream (here: synthesises it), every rank gets
its block of samples (gr_dtl_amd.sharding.scatter_stream: point-to-point RCCL sends over xGMI), decodes it, keeps
the transport blocks it owns, and the decoded bytes + block records travel back to rank 0 (gather_results).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        tools/root_fed.py [--workload c3] [--frames-per-gpu 65536] [--steps 3] [--verify]

Rank 0 prints one JSON line: times of the three phases (max over ranks), throughput with and without the exchange.
--verify: rank 0 also decodes the whole stream with one receiver and compares the bytes (small sizes).
DTL_ROOT_FED_GLOO=1: gloo backend with host staging (rehearsal on a box whose ranks share one GPU)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--frames-per-gpu", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--verify", action="store_true")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    import bench
    import gr_dtl_amd as dtl
    from gr_dtl_amd.sharding import combine, gather_results, owned_payload, scatter_stream
    gloo = os.environ.get("DTL_ROOT_FED_GLOO") == "1"
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if gloo:
        dist.init_process_group(backend="gloo")
    else:
        dist.init_process_group(backend="nccl", device_id=device)
    rank, world = dist.get_rank(), dist.get_world_size()
    w = bench.WORKLOADS[args.workload]
    cfg = bench.build_config(dtl, w)
    rx = dtl.ofdm_receiver(cfg)
    meta = torch.zeros(2, dtype=torch.int64, device="cpu" if gloo else device)
    x = None
    if rank == 0:
        pieces, n_frames = [], 0      # one piece per rank's share: the transmitter's 32-bit offsets end at ~256 k C3 frames
        for r in range(world):
            xr, nf, _, _ = bench.synth_stream(dtl, cfg, w, args.frames_per_gpu, device, seed=r, lead=(r == 0), tail=(r == world - 1))
            pieces.append(xr)
            n_frames += nf
        x = torch.cat(pieces) if world > 1 else pieces[0]
        del pieces
        meta[0], meta[1] = n_frames, x.numel()
    dist.broadcast(meta, 0)
    n_frames, total = int(meta[0]), int(meta[1])
    exch_dev = None if gloo else device
    t_scatter, t_decode, t_gather, got = [], [], [], None
    for _ in range(args.steps):
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        src = x.cpu() if (gloo and rank == 0) else x
        mine, plan = scatter_stream(src, n_frames, rx.frame_samples, 100, total, dist, device=exch_dev)
        mine = mine.to(device)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        res = rx.work(mine)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        data, tbs = owned_payload(res, plan, on_device=not gloo)
        got, recs = gather_results(data, tbs, dist, device=exch_dev)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        t_scatter.append(t1 - t0); t_decode.append(t2 - t1); t_gather.append(t3 - t2)
        del mine
    cdev = None if gloo else device
    best = int(np.argmin(np.array(t_scatter) + np.array(t_decode) + np.array(t_gather)))
    (ts, _), (td, _), (tg, _) = (combine(t[best], [0], dist, cdev) for t in (t_scatter, t_decode, t_gather))
    if rank == 0:
        line = {"workload": args.workload, "n_gpus": world, "frames": n_frames, "samples": total,
                "scatter_ms": round(ts * 1e3, 3), "decode_ms": round(td * 1e3, 3), "gather_ms": round(tg * 1e3, 3),
                "msamples_per_s_decode_only": round(total / td / 1e6, 1),
                "msamples_per_s_with_exchange": round(total / (ts + td + tg) / 1e6, 1),
                "payload_bytes": int(got.numel() if hasattr(got, "numel") else got.size), "blocks": int(len(recs)),
                "backend": "gloo (host staged rehearsal)" if gloo else "nccl (RCCL)"}
        if args.verify:
            whole = dtl.ofdm_receiver(cfg).work(x)
            got_np = got.cpu().numpy() if hasattr(got, "cpu") else got
            line["matches_single_receiver"] = bool(np.array_equal(got_np, whole.payload.cpu().numpy()))
        print(json.dumps(line))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
