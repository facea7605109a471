#!/usr/bin/env python3
"""Stability check: many steps of the default workload, alternating batch calls, streamed calls and resets; device
memory in use and the result counters must stay put."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, gr_dtl_amd as dtl
w = bench.WORKLOADS["c3"]; cfg = bench.build_config(dtl, w); rx = dtl.ofdm_receiver(cfg)
dev = torch.device("cuda:0")
x, nf, nu, user = bench.synth_stream(dtl, cfg, w, 16384, dev, seed=0)
ref = rx.work(x)
want = (len(ref.tbs), int(ref.payload.numel()), int((ref.tbs["crc_ok"] == 1).sum()))
free0 = None
n = x.numel()
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    if it % 3 == 0:
        r = rx.work(x)
        got = (len(r.tbs), int(r.payload.numel()), int((r.tbs["crc_ok"] == 1).sum()))
    else:
        cuts = [0, n // 3 + 17 * it, 2 * n // 3 + 5 * it, n]
        tb = pay = ok = 0
        for i in range(3):
            r = rx.feed(x[cuts[i]:cuts[i + 1]], final=(i == 2))
            tb += len(r.tbs); pay += int(r.payload.numel()); ok += int((r.tbs["crc_ok"] == 1).sum())
        got = (tb, pay, ok)
    assert got == want, (it, got, want)
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if it == 5:
        free0 = free
    if it > 5:
        assert free0 - free < 64 << 20, ("device memory keeps growing", it, free0, free)
print("soak ok:", want, "free MiB", free >> 20)
