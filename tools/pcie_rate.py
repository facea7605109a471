#!/usr/bin/env python3
"""PCIe-inclusive rate of the receive path (never the bench's `value`): the C3 stream starts in PINNED
HOST memory instead of HBM.
  serial     : one host->device copy of the whole batch, then one work() call
  overlapped : the batch is cut into slices; slice k+1 is copied on a second HIP stream while slice k is decoded by
               feed() (the streaming entry point keeps all receiver state across the slices)
Prints one JSON document."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import gr_dtl_amd as dtl
    dev = torch.device("cuda", 0)
    w = bench.WORKLOADS["c3"]
    cfg = bench.build_config(dtl, w)
    rx = dtl.ofdm_receiver(cfg)
    x, n_frames, _, _ = bench.synth_stream(dtl, cfg, w, int(os.environ.get("FRAMES", w["frames"])), dev, 0)
    n = x.numel()
    host = torch.empty(n, dtype=torch.complex64, pin_memory=True)
    host.copy_(x)
    ref = rx.work(x, copy_outputs=False)
    del x
    torch.cuda.synchronize()
    out = {"samples": n, "frames": n_frames}

    # ---- serial
    times = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        xd = host.to(dev, non_blocking=True)
        res = rx.work(xd, copy_outputs=False)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        del xd
    out["serial_ms"] = round(min(times) * 1e3, 2)
    out["serial_msamples_per_s"] = round(n / min(times) / 1e6, 1)
    assert res.n_payload_bytes == ref.n_payload_bytes

    # ---- overlapped: double-buffered slices
    n_slices = int(os.environ.get("SLICES", "8"))
    bounds = [n * i // n_slices for i in range(n_slices + 1)]
    copy_stream = torch.cuda.Stream()
    bufs = [torch.empty(bounds[1] - bounds[0] + n_slices, dtype=torch.complex64, device=dev) for _ in range(2)]
    ready = [torch.cuda.Event() for _ in range(2)]
    free = [torch.cuda.Event() for _ in range(2)]
    times = []
    for _ in range(3):
        rx.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        total = 0

        def enqueue_copy(k):
            b = k % 2
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(free[b])
                bufs[b][:bounds[k + 1] - bounds[k]].copy_(host[bounds[k]:bounds[k + 1]], non_blocking=True)
                ready[b].record(copy_stream)
        for b in range(2):
            free[b].record(torch.cuda.current_stream())
        enqueue_copy(0)
        for k in range(n_slices):
            if k + 1 < n_slices:
                enqueue_copy(k + 1)
            b = k % 2
            torch.cuda.current_stream().wait_event(ready[b])
            r = rx.feed(bufs[b][:bounds[k + 1] - bounds[k]], final=(k == n_slices - 1), copy_outputs=False)
            free[b].record(torch.cuda.current_stream())
            total += r.n_payload_bytes
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    out["slices"] = n_slices
    out["overlapped_ms"] = round(min(times) * 1e3, 2)
    out["overlapped_msamples_per_s"] = round(n / min(times) / 1e6, 1)
    out["payload_bytes_match_batch"] = bool(total == ref.n_payload_bytes)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
