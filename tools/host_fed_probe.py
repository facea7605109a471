#!/usr/bin/env python3
"""Timeline of host-fed operation (bench.host_fed_run with stopwatches): per slice the copy's start / end on the copy stream
(events) and the wall-clock interval of feed().  Shows what keeps the rate below the link's."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import gr_dtl_amd as dtl
    n_slices = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device("cuda", 0)
    w = dict(bench.WORKLOADS["c3"])
    cfg = bench.build_config(dtl, w)
    rx = dtl.ofdm_receiver(cfg)
    x, _, _, _ = bench.synth_stream(dtl, cfg, w, w["frames"], dev, seed=0)
    print(bench.host_fed_record(dtl, rx, x, n_slices))
    n = x.numel()
    host = torch.empty(n, dtype=torch.complex64, pin_memory=True)
    host.copy_(x)
    cuts = [(n * k // n_slices) & ~4095 for k in range(n_slices)] + [n]
    sl = [(cuts[k], cuts[k + 1]) for k in range(n_slices)]
    biggest = max(b - a for a, b in sl)
    bufs = [torch.empty(biggest, dtype=torch.complex64, device=dev) for _ in range(2)]
    cs = torch.cuda.Stream(device=dev)
    out_host = torch.empty(64 << 20, dtype=torch.uint8, pin_memory=True)
    os_ = torch.cuda.Stream(device=dev)
    for rep in range(4):
        rx.reset()
        rows = []
        t0 = time.perf_counter()
        bench.host_fed_run(rx, host, sl, bufs, cs, out_host, timeline=rows, out_stream=os_ if rep >= 2 else None)
        total = time.perf_counter() - t0
        if rep in (1, 3):
            print("bytes home on %s" % ("their own stream" if rep >= 2 else "the decode stream"))
            print("slice  copy_start copy_end | feed_start feed_end  (ms)   host stages of the call")
            for k, r in enumerate(rows):
                print("%3d   %8.2f %8.2f | %8.2f %8.2f   %s" % (k, r[0], r[1], r[2], r[3], " ".join("%.2f" % v for v in r[4])))
            print("total %.2f ms for %d samples = %.1f GB/s" % (total * 1e3, n, n * 8 / total / 1e9))


if __name__ == "__main__":
    main()
