#!/usr/bin/env python3
"""Loop an adaptive-OFDM link on one MI355X: GPU transmitter -> AWGN -> GPU receiver, fed to the receiver in
arbitrary slices the way a GNU Radio scheduler would, with the reference's telemetry shapes on the way out.

    python examples/loopback.py [--snr 24 --constellation 4 --bytes 200000]   (the defaults: a link that works;
    --snr 12 shows what the monitor reports when synchronisation starts to fail)
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fft", type=int, default=64)
    ap.add_argument("--constellation", type=int, default=4, help="1 BPSK, 2 QPSK, 3 8PSK, 4 16QAM")
    ap.add_argument("--snr", type=float, default=24.0, help="Es/N0 of a unit-energy carrier, dB")
    ap.add_argument("--bytes", type=int, default=200000)
    ap.add_argument("--slice", type=int, default=50000, help="samples per feed() call")
    args = ap.parse_args()
    import torch
    import gr_dtl_amd as dtl
    from gr_dtl_amd.ofdm_transmitter import ofdm_transmitter
    dev = torch.device("cuda", 0)
    cfg = dtl.ofdm_adaptive_rx_config() if args.fft == 64 else dtl.make_scaled_rx_config(args.fft)
    cfg.fec_codes = (("fec_1", "n_0300_k_0152_gap_03.alist"),)
    C = dtl.constellation_type_t
    cfg.mcs = ((-100000, (C.BPSK, "fec_1")), (11, (C.QPSK, "fec_1")), (16, (C.PSK8, "fec_1")), (21, (C.QAM16, "fec_1")))
    cfg.initial_mcs_id = 1
    tx, rx = ofdm_transmitter(cfg), dtl.ofdm_receiver(cfg)
    user = torch.randint(0, 256, (args.bytes,), dtype=torch.uint8, device=dev)
    iq, hdrs, _ = tx.work_device(user, args.constellation, "fec_1")
    x = torch.cat([torch.zeros(100, dtype=torch.complex64, device=dev), iq.reshape(-1),
                   torch.zeros(tx.frame_samples, dtype=torch.complex64, device=dev)])
    nv = args.fft / 10 ** (args.snr / 10)
    x = x + torch.view_as_complex(torch.randn((x.numel(), 2), device=dev) * float(np.sqrt(nv / 2)))
    probe, got, frames, tbs_ok, tbs = dtl.monitor_probe(), [], 0, 0, 0
    for lo in range(0, x.numel(), args.slice):
        hi = min(lo + args.slice, x.numel())
        res = rx.feed(x[lo:hi], final=(hi == x.numel()))
        got.append(res.payload)
        frames += res.n_payload_frames
        for m in probe.dec_msgs(res, code_n=300, code_k=152):
            tbs, tbs_ok = tbs + 1, m["crc_ok_count"]
        for m in probe.eq_msgs(res)[-1:]:
            last_eq = m
    got = torch.cat(got)
    print("sent %d frames, %d user bytes; decoded %d frames, %d/%d transport blocks CRC-ok, %d bytes delivered" % (
        len(hdrs), args.bytes, frames, tbs_ok, tbs, got.numel()))
    if frames:
        print("last frame: SNR estimate %.1f dB, feedback MCS %d, lost-frame rate %.2f %%" % (
            last_eq["estimated_snr_tag_key"], last_eq["constellation_key"], last_eq["lost_frames_rate"]))
    if got.numel() == user.numel():
        print("payload identical:", bool(torch.equal(got, user)))


if __name__ == "__main__":
    main()
