import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench, gr_dtl_amd as dtl
w = bench.WORKLOADS["c3"]; cfg = bench.build_config(dtl, w); rx = dtl.ofdm_receiver(cfg)
x, nf, nu, user = bench.synth_stream(dtl, cfg, w, 8192, torch.device("cuda:0"), seed=0)
r = rx.work(x, want=("cw",))
n = r.cw_nit.cpu().numpy()
print("codewords", n.size, "hist", np.bincount(n, minlength=16).tolist())
