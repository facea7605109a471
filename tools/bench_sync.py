#!/usr/bin/env python3
"""The sync kernel alone on a C3-shaped stream (run it under rocprofv3 --kernel-trace --stats for the kernel's duration;
prints the wall time of dtl_sync_sc, which includes the host's plateau detector)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import gr_dtl_amd as dtl

w = dict(bench.WORKLOADS["c3"])
cfg = bench.build_config(dtl, w)
dev = torch.device("cuda", 0)
frames = int(os.environ.get("FRAMES", "16384"))
x, nf, _, _ = bench.synth_stream(dtl, cfg, w, frames, dev, seed=0)
L = dtl._lib.lib()
L.dtl_sync_sc.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                          C.POINTER(C.c_int64), C.c_void_p]
cap = 4 * frames + 1024
trig = np.zeros(cap, np.int64)
ang = np.zeros(cap, np.float32)
n_out = C.c_int64(0)
for it in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = L.dtl_sync_sc(x.data_ptr(), x.numel(), cfg.fft_len, cfg.cp_len, C.c_float(cfg.sync_threshold), None, trig.ctypes.data,
                       ang.ctypes.data, cap, C.byref(n_out), None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("rc %d  %d samples  %d triggers  %.3f ms per call (kernel + host)  %.0f GB/s" % (rc, x.numel(), n_out.value, dt * 1e3, x.numel() * 8 / dt / 1e9))
