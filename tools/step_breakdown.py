#!/usr/bin/env python3
"""Where one C3 bench step spends its wall-clock: C-ABI call vs Python wrapper vs host stages."""
import os, sys, time
import ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import gr_dtl_amd as dtl
from gr_dtl_amd import _lib
from gr_dtl_amd.ofdm_receiver import _Result

w = bench.WORKLOADS["c3"]
cfg = bench.build_config(dtl, w)
rx = dtl.ofdm_receiver(cfg)
dev = torch.device("cuda", 0)
x, n_frames, _, _ = bench.synth_stream(dtl, cfg, w, int(os.environ.get("FRAMES", "65536")), dev, 0)
torch.cuda.synchronize()
for _ in range(2):
    rx.work(x, copy_outputs=False)
L = _lib.lib()
st = _lib.current_stream_ptr()
for it in range(3):
    res = _Result()
    t0 = time.perf_counter()
    L.dtl_rx_process(rx._h, x.data_ptr(), x.numel(), 0, st, C.byref(res))
    t1 = time.perf_counter()
    r = rx.work(x, copy_outputs=False)
    t2 = time.perf_counter()
    print("C call %.2f ms | work() %.2f ms | stages %s | sum %.2f" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3,
          ["%.2f" % s for s in r.stage_ms], sum(r.stage_ms[:6])))
