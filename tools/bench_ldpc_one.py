"""Single-point K10 run for profiling: python tools/bench_ldpc_one.py <ncw> <esn0_db> <reps>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gr_dtl_amd as g
ncw = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
esn0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dec = g.ldpc_dec("n_0300_k_0152_gap_03.alist")
n, k = dec.get_n(), dec.get_k()
dev = torch.device("cuda:0")
sig2 = 10 ** (-esn0 / 10) / 2
gen = torch.Generator(device=dev); gen.manual_seed(1)
y = -1.0 + torch.randn((ncw, n), device=dev, generator=gen) * sig2 ** 0.5
llr = (2 * y / sig2).float()
bits = torch.empty((ncw, k), dtype=torch.uint8, device=dev); nit = torch.empty((ncw,), dtype=torch.int32, device=dev)
dec.decode_batch(llr, bits, nit); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    dec.decode_batch(llr, bits, nit)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
its = nit.float().mean().item()
print(f"ncw={ncw} EsN0={esn0} avg_nit={its:.3f} {ms:.3f} ms {ncw*its/ms/1e3:.2f} Mcw-it/s {ncw/ms/1e3:.2f} Mcw/s")
