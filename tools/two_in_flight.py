#!/usr/bin/env python3
"""Experiment: K C3 steps with one receiver (sequential) vs two receivers on two HIP streams driven by two host
threads (the host phases and the sync kernel of one batch overlap the decoder kernels of the other)."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import gr_dtl_amd as dtl

dev = torch.device("cuda", 0)
w = bench.WORKLOADS["c3"]
cfg = bench.build_config(dtl, w)
x, n_frames, _, _ = bench.synth_stream(dtl, cfg, w, w["frames"], dev, 0)
K = int(os.environ.get("STEPS", "12"))
rxs = [dtl.ofdm_receiver(cfg) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
for rx, s in zip(rxs, streams):
    with torch.cuda.stream(s):
        for _ in range(2):
            rx.work(x, copy_outputs=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    ref = rxs[0].work(x, copy_outputs=False)
torch.cuda.synchronize()
t_seq = time.perf_counter() - t0
got = [None, None]

def run(i):
    with torch.cuda.stream(streams[i]):
        for _ in range(K // 2):
            got[i] = rxs[i].work(x, copy_outputs=False)

torch.cuda.synchronize()
t0 = time.perf_counter()
ths = [threading.Thread(target=run, args=(i,)) for i in range(2)]
for t in ths: t.start()
for t in ths: t.join()
torch.cuda.synchronize()
t_par = time.perf_counter() - t0
print("sequential %.2f ms/step, two in flight %.2f ms/step, payload bytes %d / %d / %d" % (
    t_seq / K * 1e3, t_par / K * 1e3, ref.n_payload_bytes, got[0].n_payload_bytes, got[1].n_payload_bytes))
