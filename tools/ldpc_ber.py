#!/usr/bin/env python3
"""Decoder-level BER / FER versus Eb/N0 of the (300,152) code in both arithmetics (gr_exact = the reference decoder's
operations, fast = the production arithmetic), BPSK over AWGN, RANDOM codewords (the all-zero codeword flatters the
exact form: its NaN collapse decides 0 for every bit).  Same noise for both modes.  One JSON document.
The 0.1 dB criterion of BASELINE.json is read off these curves: at a given BER, the Eb/N0 the two modes need."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ncw", type=int, default=1 << 18)
    ap.add_argument("--ebn0", default="0:5.01:0.25")
    ap.add_argument("--alist", default="n_0300_k_0152_gap_03.alist")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    import gr_dtl_amd as dtl
    dev = torch.device("cuda", 0)
    decs = {a: dtl.ldpc_dec(args.alist, arith=a) for a in ("gr_exact", "fast")}
    n, k = decs["fast"].get_n(), decs["fast"].get_k()
    rng = np.random.default_rng(7)
    data = rng.integers(0, 2, (args.ncw, k), dtype=np.uint8)
    cws = torch.from_numpy(decs["fast"].encode_batch(data)).to(dev)
    data_d = torch.from_numpy(data).to(dev)
    tx = cws.float() * 2 - 1                                    # bit 1 -> +1 (positive soft value = 1)
    lo, hi, st = (float(v) for v in args.ebn0.split(":"))
    rate = k / n
    gen = torch.Generator(device=dev)
    points = []
    e = lo
    while e < hi:
        esn0 = e + 10 * np.log10(rate)
        sig2 = 10 ** (-esn0 / 10) / 2
        gen.manual_seed(int(round(e * 100)) + 12345)
        y = tx + torch.randn(tx.shape, device=dev, generator=gen) * sig2 ** 0.5
        llr = (2 * y / sig2).float()
        rec = {"ebn0_db": round(e, 3), "esn0_db": round(float(esn0), 3), "codewords": args.ncw, "info_bits": args.ncw * k}
        for a, dec in decs.items():
            bits, nit = dec.decode_batch(llr)
            err = bits != data_d
            rec[a] = {"bit_errors": int(err.sum().item()), "frame_errors": int(err.any(1).sum().item()),
                      "ber": float(err.float().mean().item()), "fer": float(err.any(1).float().mean().item()),
                      "iterations_per_codeword": round(float(nit.float().mean().item()), 3)}
        points.append(rec)
        print("Eb/N0 %5.2f dB  gr_exact BER %.3e FER %.3e | fast BER %.3e FER %.3e" % (
            e, rec["gr_exact"]["ber"], rec["gr_exact"]["fer"], rec["fast"]["ber"], rec["fast"]["fer"]), file=sys.stderr, flush=True)
        e += st
    doc = {"code": args.alist, "n": n, "k": k, "modulation": "BPSK, AWGN, random codewords", "max_iterations": 15,
           "points": points}
    text = json.dumps(doc, indent=1)
    if args.out:
        open(args.out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
