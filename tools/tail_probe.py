#!/usr/bin/env python3
"""Which layer issues the small device copies seen between two steps in the kernel trace?  Phase A: the C entry point
called directly three times; a marker kernel; phase B: the Python wrapper three times.  Run under
rocprofv3 --kernel-trace and list the kernels between the decoder launches (tools/timeline.py style)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import gr_dtl_amd as dtl
from gr_dtl_amd import _lib
from gr_dtl_amd.ofdm_receiver import _Result

w = dict(bench.WORKLOADS["c3"])
cfg = bench.build_config(dtl, w)
rx = dtl.ofdm_receiver(cfg)
dev = torch.device("cuda", 0)
x, n_frames, _, _ = bench.synth_stream(dtl, cfg, w, 16384, dev, 0)
torch.cuda.synchronize()
rx.work(x, copy_outputs=False)
L = _lib.lib()
st = _lib.current_stream_ptr()
marker = torch.zeros(1 << 20, device=dev)
torch.cuda.synchronize()
for it in range(3):
    res = _Result()
    L.dtl_rx_process(rx._h, x.data_ptr(), x.numel(), 0, st, C.byref(res))
marker.add_(1.0)          # an elementwise kernel as a separator in the trace
torch.cuda.synchronize()
for it in range(3):
    r = rx.work(x, copy_outputs=False)
marker.add_(1.0)
torch.cuda.synchronize()
