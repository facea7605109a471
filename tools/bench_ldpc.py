"""Micro-benchmark of K10 (LDPC BP) alone: codeword-iterations/s at a few operating points."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gr_dtl_amd as g

dec = g.ldpc_dec("n_0300_k_0152_gap_03.alist")
n, k = dec.get_n(), dec.get_k()
dev = torch.device("cuda:0")
ncw = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
for esn0 in (-1.0, 0.0, 1.0, 2.0, 4.0, 8.0):
    sig2 = 10 ** (-esn0 / 10) / 2
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    y = -1.0 + torch.randn((ncw, n), device=dev, generator=gen) * sig2 ** 0.5   # all-zero codeword
    llr = (2 * y / sig2).float()
    bits = torch.empty((ncw, k), dtype=torch.uint8, device=dev)
    nit = torch.empty((ncw,), dtype=torch.int32, device=dev)
    dec.decode_batch(llr, bits, nit); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        dec.decode_batch(llr, bits, nit)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    its = nit.float().mean().item()
    fer = (bits != 0).any(1).float().mean().item()
    print(f"EsN0={esn0:5.1f} dB  avg_nit={its:5.2f} FER={fer:.4f}  {ms:8.3f} ms  "
          f"{ncw/ms/1e3:8.2f} Mcw/s  {ncw*its/ms/1e3:8.2f} Mcw-it/s  llr-read {ncw*n*4/ms/1e6:7.1f} GB/s", flush=True)
