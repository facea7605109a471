#!/usr/bin/env python3
"""Per-kernel device times of one receive step of a bench workload (dtl_rx_kernel_split: HIP events around every launch of
one call), one line.  usage: tools/kernel_split.py [--workload c3] [--frames N] [--esn0 dB]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--esn0", type=float, default=None)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import bench
    import gr_dtl_amd as dtl
    dev = torch.device("cuda", 0)
    w = dict(bench.WORKLOADS[args.workload])
    if args.esn0 is not None:
        w["esn0_db"] = args.esn0
    cfg = bench.build_config(dtl, w)
    rx = dtl.ofdm_receiver(cfg)
    x, n_frames, _, _ = bench.synth_stream(dtl, cfg, w, args.frames or w["frames"], dev, seed=0)
    rx.work(x, copy_outputs=False)
    best = None
    for _ in range(args.reps):
        sp = bench.kernel_split(rx, x)
        tot = sum(v["ms"] for v in sp.values())
        if best is None or tot < best[0]:
            best = (tot, sp)
    print("%s %d frames: " % (args.workload, n_frames) + "  ".join("%s %.3f ms (%d)" % (k.split(" ")[0].replace("_kernel", ""), v["ms"], v["launches"])
                                                                  for k, v in best[1].items()) + "  | sum %.3f ms" % best[0])


if __name__ == "__main__":
    main()
