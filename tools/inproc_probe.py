#!/usr/bin/env python3
"""Is a workload slower when other receivers have run in the same process before it?  (bench.py's extras measured c5 at
15.6 ms where a process of its own measures 14.5.)  usage: inproc_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import gr_dtl_amd as dtl

dev = torch.device("cuda", 0)
barrier = lambda: torch.cuda.synchronize()


def run(name, steps):
    w = dict(bench.WORKLOADS[name])
    cfg = bench.build_config(dtl, w)
    rx = dtl.ofdm_receiver(cfg)
    x, nf, nu, _ = bench.synth_stream(dtl, cfg, w, w["frames"], dev, seed=0)
    torch.cuda.synchronize()
    e, res, *_ = bench.timed_steps(rx, x, steps, 4, barrier, pipelined=True)
    return e / steps * 1e3, rx, x


a, rx5, x5 = run("c5", 20)
print("c5 first in the process: %.3f ms" % a)
del rx5, x5
b, rx3, x3 = run("c3", 10)
print("c3: %.3f ms" % b)
c, rx5, x5 = run("c5", 20)
print("c5 with the c3 receiver and stream alive: %.3f ms" % c)
del rx3, x3
torch.cuda.empty_cache()
d, rx5b, x5b = run("c5", 20)
print("c5 after they were freed: %.3f ms" % d)
