"""Diagnostic (round 4): where do the product's and the oracle's equalised symbols differ in the sign of a zero?
For one seed of tests/test_fuzz_gpu.py's rotating-carrier-set scenarios: the mismatching (frame, index) positions, the
oracle's FFT value there, the product's block entry point on the oracle's rows, and the carrier's whole symbol sequence.
(Answer: the oracle composed complex values as re + im * I, which turns a real part of -0 into +0; fixed with CMPLXF.)"""
import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import conftest
import gr_dtl_amd as dtl
from oracle import oracle
import test_fuzz_gpu as tf
from test_stages_gpu import _frames_in_frequency_domain
from helpers import oracle_feedback
cuda = torch.device("cuda", 0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 14
rng = np.random.default_rng(456000 + seed)
pcfg, ocfg, code, x, info = tf._scenario(oracle, dtl, rng, rotating_sets=True)
rx = dtl.ofdm_receiver(pcfg)
res = rx.work(torch.from_numpy(x).to(cuda), want=("eq_soft", "eq_hard", "taps", "soft_syms"))
ref = oracle.rx_stream(ocfg, code, x, feedback=oracle_feedback(oracle, pcfg, rx), max_frames=x.size // 64 + 16,
                       want=("frames", "trig", "angle", "eq_hard", "eq_soft", "soft_syms", "taps"))
eq_soft = res.eq_soft.cpu().numpy()
N, L, h = ocfg.fft_len, ocfg.frame_length, ocfg.header_len
pi = 0
bad = []
for i, rf in enumerate(ref["frames"]):
    if not rf["header_ok"]:
        continue
    g = np.ascontiguousarray(eq_soft[pi]).reshape(-1); w = np.ascontiguousarray(ref["eq_soft"][i]).reshape(-1)
    d = np.nonzero((g.view(np.uint32).reshape(-1, 2) != w.view(np.uint32).reshape(-1, 2)).any(axis=1))[0]
    for j in d:
        bad.append((i, pi, int(j)))
    pi += 1
print("mismatches", bad)
ref2, frames = _frames_in_frequency_domain(oracle, ocfg, code, x)
by_i = {i: (f, rows) for i, f, rows in frames}
for (i, p, j) in bad[:4]:
    f, rows = by_i[i]
    sym, k = divmod(j, N)
    row = rows[2 + h + sym]
    print("frame", i, "sym", sym, "carrier", k, "oracle FFT value", row[k], np.signbit(row[k].real), np.signbit(row[k].imag), "carr_off", f["carr_offset"])
    taps = torch.from_numpy(np.ascontiguousarray(ref["taps"][i])[None]).to(cuda)
    hard, soft, syms, stats = rx.equalize_frames(torch.from_numpy(rows[None, 2 + h:]).to(cuda), taps, [f["hdr"]["cnst"]], [f["carr_offset"]], [f["n_payload_syms"]])
    s = soft.cpu().numpy().reshape(-1)[j]
    print("   product stage on the oracle's rows:", s, np.signbit(s.real), " product chain:", eq_soft[p].reshape(-1)[j], " oracle:", ref["eq_soft"][i].reshape(-1)[j])
    for sy in range(L):
        jj = sy * N + k
        print("     sym", sy, "row", rows[2 + h + sy][k], "| product", eq_soft[p].reshape(-1)[jj], res.eq_hard.cpu().numpy()[p].reshape(-1)[jj],
              "| oracle", ref["eq_soft"][i].reshape(-1)[jj], ref["eq_hard"][i].reshape(-1)[jj], "| tap0", ref["taps"][i][k])
    # the product's own FFT of the same samples
    mixed = oracle.cfo_mix(ocfg, x, ref["angle"])
    P, cp = N + ocfg.cp_len, ocfg.cp_len
    t = f["trigger"]; s_idx = 2 + h + sym
    seg = mixed[t + s_idx * P + cp:t + s_idx * P + cp + N].astype(np.complex64)
    from gr_dtl_amd import stages
    mine = stages.fft_batch(torch.from_numpy(seg[None]).to(cuda)).cpu().numpy()[0]
    print("   product FFT of the oracle's mixed samples:", mine[k], np.signbit(mine[k].real), np.signbit(mine[k].imag))
