#!/usr/bin/env python3
"""Does a second receiver handle on its own stream, fed alternate batches from a second host thread, fill the device
idle of the single-handle step?  Prints ms per step for depth 1, 2, 3 on the bench's C3 stream."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gr_dtl_amd as dtl
import bench

device = torch.device("cuda", 0)
torch.cuda.set_device(0)
w = dict(bench.WORKLOADS["c3"])
cfg = bench.build_config(dtl, w)
cfg.arith = os.environ.get("ARITH", "gr_exact")
x, n_frames, n_user, _ = bench.synth_stream(dtl, cfg, w, int(os.environ.get("FRAMES", w["frames"])), device, seed=0)
torch.cuda.synchronize()
steps = int(os.environ.get("STEPS", "12"))
for depth in (1, 2, 3, 1, 2):
    rxs = [dtl.ofdm_receiver(cfg) for _ in range(depth)]
    streams = [torch.cuda.Stream(device=device) for _ in range(depth)]
    last = [None] * depth
    for j in range(depth):
        with torch.cuda.stream(streams[j]):
            for _ in range(2):
                last[j] = rxs[j].work(x, copy_outputs=False)
    torch.cuda.synchronize()

    def worker(j):
        torch.cuda.set_device(0)
        with torch.cuda.stream(streams[j]):
            for _ in range(j, steps, depth):
                last[j] = rxs[j].work(x, copy_outputs=False)

    t0 = time.perf_counter()
    ts = [threading.Thread(target=worker, args=(j,)) for j in range(depth)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = all(int(r.n_payload_bytes) == int(last[0].n_payload_bytes) for r in last)
    ks = [rxs[j].kernel_ms(0) for j in range(depth)]
    print("depth %d: %.3f ms per step (%d steps), payload bytes %d, all handles equal: %s, last-call kernel ms %s"
          % (depth, dt / steps * 1e3, steps, int(last[0].n_payload_bytes), same, ks), flush=True)
    del rxs
    torch.cuda.empty_cache()
