#!/usr/bin/env python3
"""Is the decoder's per-codeword fixed cost (codewords that need no iteration) memory latency?  The same 2^20 clean
codewords (a) spread over 1.26 GB of soft bits (HBM) and (b) all reading the same 1.2 MB (L2-resident), through the
gather entry point; then (c) the spread set again with every descriptor pointing at its own soft bits but the batch
decoded in the receive chain's mix (54 % no iteration, the rest ~1.4) for the record."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import gr_dtl_amd as dtl
from gr_dtl_amd import _lib

dev = torch.device("cuda", 0)
dec = dtl.ldpc_dec("n_0300_k_0152_gap_03.alist", arith=os.environ.get("ARITH", "gr_exact"))
n, k = dec.get_n(), dec.get_k()
ncw = 1 << 20
g = torch.Generator(device=dev)
g.manual_seed(3)


def timed(llr, desc, label):
    bits = torch.zeros((ncw * k + 7) // 8 + 64, dtype=torch.uint8, device=dev) if False else torch.empty((ncw, k), dtype=torch.uint8, device=dev)
    nit = torch.empty(ncw, dtype=torch.int32, device=dev)
    dec.decode_gather(llr, desc, bits, nit)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dec.decode_gather(llr, desc, bits, nit)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("%-46s %7.3f ms  %7.1f Mcw/s  %.2f it/cw  %6.0f GB/s of soft bits" % (label, ms, ncw / ms / 1e3, nit.float().mean().item(), ncw * n * 4 / ms / 1e6))


def descs(offsets):
    d = np.zeros(ncw, dtype=np.dtype([("llr_off0", "<u4"), ("llr_off1", "<u4"), ("out_off", "<u4"), ("len0", "<u2"), ("len1", "<u2"),
                                      ("k_used", "<u2"), ("pad", "<u2")]))
    d["llr_off0"] = offsets
    d["out_off"] = np.arange(ncw, dtype=np.uint32) * k
    d["len0"] = n
    d["k_used"] = k
    return torch.from_numpy(d.view(np.uint8).reshape(-1).copy()).to(dev)


for esn0 in (8.0, 2.0):
    sig2 = 10 ** (-esn0 / 10) / 2
    y = -1.0 + torch.randn((ncw, n), device=dev, generator=g) * sig2 ** 0.5
    llr = (2 * y / sig2).float().reshape(-1)
    del y
    spread = np.arange(ncw, dtype=np.uint64) * n
    timed(llr, descs(spread.astype(np.uint32)), "Es/N0 %.0f dB, soft bits spread over %.2f GB" % (esn0, ncw * n * 4 / 1e9))
    timed(llr, descs((spread % (1024 * n)).astype(np.uint32)), "Es/N0 %.0f dB, every read inside 1.2 MB (L2)" % esn0)
    perm = np.random.default_rng(1).permutation(ncw).astype(np.uint64)
    timed(llr, descs((perm * n).astype(np.uint32)), "Es/N0 %.0f dB, spread, codewords in random order" % esn0)
