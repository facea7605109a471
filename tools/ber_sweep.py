#!/usr/bin/env python3
"""BER / block-error-rate versus Es/N0 of the product receive chain on a mixed-MCS AWGN stream
(BASELINE.json configs[4]: per-frame adaptive MCS, 1024-point FFT, LDPC).  Product TX (numpy) -> AWGN -> product RX
(HIP, through the C ABI).  One JSON document on stdout / --out.  The same streams are decoded by the CPU oracle in
tests/test_ber_sweep_gpu.py, which asserts identical error counts."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_config(dtl, fft_len, frame_length, threshold, sync_correct=True, arith="gr_exact"):
    cfg = dtl.ofdm_adaptive_rx_config(frame_length=frame_length) if fft_len == 64 else \
        dtl.make_scaled_rx_config(fft_len, frame_length=frame_length)
    cfg.sync_threshold = threshold
    cfg.use_sync_correct = sync_correct
    cfg.arith = arith
    cfg.fec_codes = (("fec_1", "n_0300_k_0152_gap_03.alist"),)
    C = dtl.constellation_type_t
    cfg.mcs = ((-100000, (C.BPSK, "fec_1")), (11, (C.QPSK, "fec_1")), (16, (C.PSK8, "fec_1")), (21, (C.QAM16, "fec_1")))
    cfg.initial_mcs_id = 1
    return cfg


def sweep_point(dtl, cfg, snr_db, frames_per_mcs, seed, device, gap=0):
    """Time-domain SNR (signal power of the payload symbols over noise power per sample), one stream that cycles
    BPSK -> QPSK -> 8PSK -> 16QAM twice."""
    import torch
    from gr_dtl_amd.link_sim import add_awgn, bit_errors, mixed_mcs_stream
    from gr_dtl_amd.ofdm_transmitter import ofdm_transmitter
    tx = ofdm_transmitter(cfg)
    rng = np.random.default_rng(seed)
    segs = []
    for _ in range(2):
        for cnst in (1, 2, 3, 4):
            frame_bytes = tx.frame_capacity * cnst // 8
            segs.append((cnst, int(frames_per_mcs * frame_bytes * 0.49)))
    x, sent, _ = mixed_mcs_stream(tx, segs, rng, gap=gap)
    # noise variance per time-domain sample = fft_len / 10^(snr/10), i.e. Es/N0 of a unit-energy subcarrier symbol
    # (QPSK carriers carry the reference's extra 0.5 amplitude factor and sit 6 dB lower)
    nv = cfg.fft_len / 10 ** (snr_db / 10)
    y = add_awgn(x, nv, rng)
    rx = dtl.ofdm_receiver(cfg)
    res = rx.work(torch.from_numpy(y).to(device))
    acc = bit_errors(sent, res, 256)
    acc.update(snr_db=snr_db, frames_decoded=int(res.n_payload_frames), tbs_seen=int(len(res.tbs)),
               tbs_crc_ok=int((res.tbs["crc_ok"] == 1).sum()),
               avg_bp_iterations=float(res.tbs["avg_it"].mean()) if len(res.tbs) else None, samples=int(y.size))
    return acc, y, sent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fft", type=int, default=1024)
    ap.add_argument("--frame-length", type=int, default=10)
    ap.add_argument("--threshold", type=float, default=0.95)
    ap.add_argument("--arith", default="gr_exact", choices=("gr_exact", "fast"))
    ap.add_argument("--snr", default="14:32:2", help="start:stop:step in dB (stop exclusive)")
    ap.add_argument("--frames-per-mcs", type=int, default=24)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--gap", type=int, default=0, help="zero samples between frames (paced TX); implies no frame_detect_bb")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    import gr_dtl_amd as dtl
    if not torch.cuda.is_available():
        raise SystemExit("ber_sweep.py needs a HIP device: the product path has no CPU fallback")
    device = torch.device("cuda", 0)
    cfg = make_config(dtl, args.fft, args.frame_length, args.threshold, sync_correct=(args.gap == 0), arith=args.arith)
    lo, hi, st = (float(v) for v in args.snr.split(":"))
    points = []
    snr = lo
    while snr < hi - 1e-9:
        acc, _, _ = sweep_point(dtl, cfg, snr, args.frames_per_mcs, args.seed, device, gap=args.gap)
        points.append(acc)
        print("Es/N0 %5.1f dB  BER %.3e  BLER %.3f  frames %d  TBs ok %d/%d" % (
            snr, acc["ber"], acc["bler"], acc["frames_decoded"], acc["tbs_crc_ok"], acc["blocks_sent"]), file=sys.stderr)
        snr += st
    doc = dict(workload="mixed-MCS (BPSK/QPSK/8PSK/16QAM cycling) AWGN stream, %d-point FFT, LDPC(300,152), "
                        "%d frames per MCS segment" % (args.fft, args.frames_per_mcs),
               snr_definition="noise variance per time-domain sample = fft_len / 10^(snr/10)",
               sync_threshold=args.threshold, arith=args.arith, frame_length=args.frame_length, seed=args.seed, gap=args.gap,
               use_sync_correct=(args.gap == 0), points=points)
    text = json.dumps(doc, indent=1)
    if args.out:
        open(args.out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
