#!/usr/bin/env python3
"""The decoder kernel alone (dense entry point) on 2^LOG2_NCW codewords of a chosen kind; prints the launch time.
  MODE=zero   clean codewords: the first syndrome check passes, no iteration (all-zero codeword at ESN0 dB, default 12)
  MODE=one    codewords that need EXACTLY one iteration: strong soft bits with one weak wrong one at a random position
  MODE=mix    the receive chain's mix: all-zero codeword over AWGN at ESN0 dB (2 dB: about 2 iterations per codeword)
Env: ARITH (gr_exact / fast), LOG2_NCW (20), ESN0, DTL_LDPC_CHUNK.  Used by tools/ablate.py (variants 21-26) and
tools/ldpc_one_report.sh."""
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
mode = os.environ.get("MODE", "zero")
g = torch.Generator(device=dev)
g.manual_seed(3)
if mode == "one":
    llr = (-6.0 + 0.5 * torch.randn((ncw, n), device=dev, generator=g)).float()        # all-zero codeword, strong
    pos = torch.randint(0, n, (ncw,), device=dev, generator=g)
    llr[torch.arange(ncw, device=dev), pos] = 1.0                                        # one weak wrong soft bit
else:
    esn0 = float(os.environ.get("ESN0", "12" if mode == "zero" else "2"))
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
nf = nit.float()
print("%s 2^%d codewords: %.3f ms per launch = %.3f ns per codeword, %.3f it/cw (min %d max %d), wrong bits %d"
      % (mode, ncw.bit_length() - 1, e0.elapsed_time(e1) / 3, e0.elapsed_time(e1) / 3 * 1e6 / ncw, nf.mean().item(), int(nit.min()), int(nit.max()),
         int(bits.sum().item())))
