#!/usr/bin/env python3
"""Where a step's wall time goes on the host side: Python wrapper vs the C call (sum of its stage timers) vs the
stages themselves (default bench workload, one batch in flight)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, gr_dtl_amd as dtl
w = bench.WORKLOADS["c3"]; cfg = bench.build_config(dtl, w); rx = dtl.ofdm_receiver(cfg)
dev = torch.device("cuda:0")
x, nf, nu, user = bench.synth_stream(dtl, cfg, w, w["frames"], dev, seed=0)
for _ in range(3):
    rx.work(x, copy_outputs=False)
torch.cuda.synchronize()
tw, tc, st = [], [], np.zeros(8)
t_all = time.perf_counter()
for _ in range(10):
    t0 = time.perf_counter()
    r = None
    r = rx.work(x, copy_outputs=False)
    tw.append(time.perf_counter() - t0)
    tc.append(sum(r.stage_ms[:6]))
    st += np.array(r.stage_ms)
torch.cuda.synchronize()
t_all = (time.perf_counter() - t_all) / 10
print("step %.3f ms   work() %.3f ms   C stages %.3f ms   stages %s" % (t_all * 1e3, np.mean(tw) * 1e3, np.mean(tc), np.round(st[:6] / 10, 3)))
