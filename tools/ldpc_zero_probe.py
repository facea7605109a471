#!/usr/bin/env python3
"""The decoder alone on 2^20 codewords that need no iteration (all-zero codeword at 12 dB), dense entry point; prints
the launch time.  Run under rocprofv3 --pmc by tools/ldpc_zero_pmc.sh to see what the per-codeword fixed cost is."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gr_dtl_amd as dtl

dev = torch.device("cuda", 0)
dec = dtl.ldpc_dec("n_0300_k_0152_gap_03.alist", arith=os.environ.get("ARITH", "gr_exact"))
n, k = dec.get_n(), dec.get_k()
ncw = 1 << int(os.environ.get("LOG2_NCW", "20"))
esn0 = float(os.environ.get("ESN0", "12"))
g = torch.Generator(device=dev)
g.manual_seed(3)
sig2 = 10 ** (-esn0 / 10) / 2
y = -1.0 + torch.randn((ncw, n), device=dev, generator=g) * sig2 ** 0.5
llr = (2 * y / sig2).float()
del y
bits = torch.empty((ncw, k), dtype=torch.uint8, device=dev)
nit = torch.empty(ncw, dtype=torch.int32, device=dev)
dec.decode_batch(llr, bits, nit)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    dec.decode_batch(llr, bits, nit)
e1.record()
torch.cuda.synchronize()
print("2^%d codewords: %.3f ms per launch, %.3f it/cw, wrong bits %d" % (ncw.bit_length() - 1, e0.elapsed_time(e1) / 3, nit.float().mean().item(), int(bits.sum().item())))
