#!/usr/bin/env python3
"""Do two receiver handles on two host threads and two streams overlap?  Each thread calls dtl_rx_process directly (ctypes
releases the interpreter lock for the call; nothing of the result is touched in Python), K batches of the C2 workload
each; compared with one thread doing 2 K batches.  usage: two_handles_probe.py [c2|c3] [K]"""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import gr_dtl_amd as dtl
from gr_dtl_amd import _lib
from gr_dtl_amd.ofdm_receiver import _Result

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
w = dict(bench.WORKLOADS[name])
cfg = bench.build_config(dtl, w)
dev = torch.device("cuda", 0)
x, n_frames, _, _ = bench.synth_stream(dtl, cfg, w, w["frames"], dev, 0)
torch.cuda.synchronize()
L = _lib.lib()


def run(rx, stream, k, out, idx, gate):
    res = _Result()
    sp = C.c_void_p(stream.cuda_stream)
    for _ in range(3):
        L.dtl_rx_process(rx._h, x.data_ptr(), x.numel(), 0, sp, C.byref(res))
    gate.wait()
    t0 = time.perf_counter()
    for _ in range(k):
        rc = L.dtl_rx_process(rx._h, x.data_ptr(), x.numel(), 0, sp, C.byref(res))
        assert rc == 0
    out[idx] = (time.perf_counter() - t0, int(res.n_payload_bytes))


for n_thr in (1, 2, 3):
    rxs = [dtl.ofdm_receiver(cfg) for _ in range(n_thr)]
    streams = [torch.cuda.Stream() for _ in range(n_thr)]
    out = [None] * n_thr
    gate = threading.Barrier(n_thr + 1)
    th = [threading.Thread(target=run, args=(rxs[i], streams[i], K, out, i, gate)) for i in range(n_thr)]
    for t in th:
        t.start()
    gate.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    print("%s: %d handle(s) x %d batches: %.3f ms per batch (wall %.1f ms; per thread %s), payload %s"
          % (name, n_thr, K, wall / (n_thr * K) * 1e3, wall * 1e3, ["%.1f" % (o[0] * 1e3) for o in out], out[0][1]))
