#!/usr/bin/env python3
"""Operating-point scan for the bench workloads: for each Es/N0 (and sync threshold) decode a short stream of the
workload's shape and print how many frames / transport blocks come through and how hard belief propagation works.
Used to choose the Es/N0 of bench.py's C2 / C3 lines: the lowest one at which the chain
works with the reference's default sync_threshold 0.95: >= 99 % of the frames decoded AND >= 90 % of the transport blocks CRC-ok."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--frames", type=int, default=2048)
    ap.add_argument("--snr", default="14:27:1")
    ap.add_argument("--thresholds", default="0.95")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    import bench
    import gr_dtl_amd as dtl
    device = torch.device("cuda", 0)
    lo, hi, st = (float(v) for v in args.snr.split(":"))
    rows = []
    for thr in (float(t) for t in args.thresholds.split(",")):
        snr = lo
        while snr < hi - 1e-9:
            w = dict(bench.WORKLOADS[args.workload])
            w["esn0_db"] = snr
            w["sync_threshold"] = thr
            cfg = bench.build_config(dtl, w)
            rx = dtl.ofdm_receiver(cfg)
            x, n_frames, n_user, _ = bench.synth_stream(dtl, cfg, w, args.frames, device, seed=0)
            res = rx.work(x, want=("cw",))
            nit = res.cw_nit.float() if res.cw_nit is not None else torch.zeros(1)
            ok = int((res.tbs["crc_ok"] == 1).sum())
            row = dict(workload=args.workload, esn0_db=snr, sync_threshold=thr, frames_sent=n_frames,
                       frames_decoded=int(res.n_payload_frames), tbs=int(len(res.tbs)), tbs_crc_ok=ok,
                       payload_bytes=int(res.n_payload_bytes), user_bytes_sent=int(n_user),
                       bp_it_per_cw=round(float(nit.mean()), 3), cw_zero_it=round(float((nit == 0).float().mean()), 4),
                       cw_max_it=round(float((nit == 15).float().mean()), 4))
            rows.append(row)
            print(json.dumps(row), file=sys.stderr, flush=True)
            del x, res, rx
            snr += st
    text = json.dumps(rows, indent=1)
    if args.out:
        open(args.out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
