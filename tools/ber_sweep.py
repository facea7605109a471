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


def deep_point(dtl, cfg, snr_db, frames_per_segment, n_cycles, seed, device, gap=0):
    """One point of the deep sweep: the mixed-MCS stream is framed / encoded / modulated by the product's GPU
    transmitter (dtl_tx_process: bit-identical to the oracle's TX, tests/test_tx_gpu.py), noise is drawn on the device,
    and the accounting is that of the reference's tools/ber.py (:131-133): a block that is not delivered counts with all
    its bits, a delivered block with its wrong bits (none, unless a CRC-32 lets a wrong block through).
    -> dict with the totals and the same figures per constellation."""
    import torch
    from gr_dtl_amd.ofdm_transmitter import ofdm_transmitter
    tx = ofdm_transmitter(cfg)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    parts, sent = [torch.zeros(100, dtype=torch.complex64, device=device)], []   # sent: (number mod 256, cnst, lo, hi, user)
    users = []
    frame_no = tb_no = n_frames_sent = 0
    n_code, k_code = 300, 152
    for cyc in range(n_cycles):
        for cnst in (1, 2, 3, 4):
            frame_bits = tx.frame_capacity * cnst
            ncws = 1 + frame_bits // n_code if frame_bits > n_code else 1
            per = ncws * k_code // 8 - 4                                  # user bytes per transport block
            on_air = ncws * (n_code - k_code) + (per + 4) * 8
            n_blocks = max(1, int(frames_per_segment * frame_bits / on_air))
            user = torch.randint(0, 256, (n_blocks * per,), dtype=torch.uint8, device=device, generator=g)
            iq, hdrs, _ = tx.work_device(user, cnst, "fec_1", first_frame_no=frame_no, first_tb_no=tb_no, gap=gap)
            parts.append(iq.reshape(-1))
            frame_no = (frame_no + len(hdrs)) & 0xFFF
            n_frames_sent += len(hdrs)
            u = user.cpu().numpy()
            for b in range(n_blocks):
                tb_no += 1
                sent.append((tb_no & 0xFF, cnst, len(users), b * per, (b + 1) * per))
            users.append(u)
    parts.append(torch.zeros(tx.frame_samples, dtype=torch.complex64, device=device))
    x = torch.cat(parts)
    del parts
    nv = cfg.fft_len / 10 ** (snr_db / 10)
    noise = torch.randn((x.numel(), 2), generator=g, device=device, dtype=torch.float32) * float(np.sqrt(nv / 2))
    x += torch.view_as_complex(noise)
    del noise
    rx = dtl.ofdm_receiver(cfg)
    res = rx.work(x)
    payload = res.payload.cpu().numpy()
    tbs = res.tbs
    per_c = {c: dict(bits_sent=0, bit_errors=0, blocks_sent=0, blocks_ok=0) for c in (1, 2, 3, 4)}
    for no, cnst, ui, lo, hi in sent:
        per_c[cnst]["bits_sent"] += (hi - lo) * 8
        per_c[cnst]["blocks_sent"] += 1
    delivered = np.zeros(len(sent), dtype=bool)
    wrong_bits = {c: 0 for c in (1, 2, 3, 4)}
    sent_no = np.array([t[0] for t in sent], dtype=np.int64)
    cursor = 0
    # A delivered block names itself by an 8-bit number only, and whole segments (hundreds of blocks of one constellation)
    # can be missing in front of it: among the sent blocks behind the cursor that carry this number (one in 256) the
    # first one with the delivered CONTENT is the match -- a CRC-32 has vouched for the bytes.  A block that matches
    # none within 16 number wraps counts as wrong against the first candidate.
    for t in tbs[tbs["crc_ok"] == 1]:
        nbytes = int(t["user_bits"]) // 8
        off = int(t["out_offset"])
        got = payload[off:off + nbytes]
        cand = cursor + np.nonzero(sent_no[cursor:cursor + 4096] == int(t["tb_no"]) % 256)[0]
        hit = None
        for j in cand:
            no, cnst, ui, lo, hi = sent[j]
            if not delivered[j] and hi - lo == nbytes and np.array_equal(got, users[ui][lo:hi]):
                hit = j
                break
        if hit is None:
            for j in cand:
                no, cnst, ui, lo, hi = sent[j]
                if not delivered[j] and hi - lo == nbytes:
                    wrong_bits[cnst] += int(np.unpackbits(np.bitwise_xor(got, users[ui][lo:hi])).sum())
                    hit = j
                    break
        if hit is not None:
            delivered[hit] = True
            per_c[sent[hit][1]]["blocks_ok"] += 1
            cursor = hit + 1
    for j, (no, cnst, ui, lo, hi) in enumerate(sent):
        if not delivered[j]:
            per_c[cnst]["bit_errors"] += (hi - lo) * 8
    tot = dict(bits_sent=0, bit_errors=0, blocks_sent=0, blocks_ok=0)
    for c in (1, 2, 3, 4):
        per_c[c]["bit_errors"] += wrong_bits[c]
        for k in tot:
            tot[k] += per_c[c][k]
        per_c[c]["ber"] = per_c[c]["bit_errors"] / max(1, per_c[c]["bits_sent"])
        per_c[c]["bler"] = 1.0 - per_c[c]["blocks_ok"] / max(1, per_c[c]["blocks_sent"])
    tot["ber"] = tot["bit_errors"] / max(1, tot["bits_sent"])
    tot["bler"] = 1.0 - tot["blocks_ok"] / max(1, tot["blocks_sent"])
    tot.update(snr_db=snr_db, wrong_bits_in_delivered_blocks=int(sum(wrong_bits.values())),
               frames_sent=int(n_frames_sent), frames_decoded=int(res.n_payload_frames),
               tbs_seen=int(len(tbs)), tbs_crc_ok=int((tbs["crc_ok"] == 1).sum()),
               avg_bp_iterations=float(tbs["avg_it"].mean()) if len(tbs) else None, samples=int(x.numel()),
               per_constellation={{1: "bpsk", 2: "qpsk", 3: "psk8", 4: "qam16"}[c]: per_c[c] for c in (1, 2, 3, 4)})
    return tot


def crossing_db(points, key, target, sub=None):
    """Es/N0 at which the curve first falls to `target`: log-linear interpolation between the two neighbouring points
    (a point without a single error is taken at half an error for the interpolation).  None: never reached."""
    def val(p):
        q = p["per_constellation"][sub] if sub else p
        return q[key] if q[key] > 0 else 0.5 / max(1, q["bits_sent"])
    for i, p in enumerate(points):
        if val(p) <= target:
            if i == 0:
                return float(p["snr_db"])
            q = points[i - 1]
            a, b = np.log10(val(q)), np.log10(val(p))
            return float(q["snr_db"] + (p["snr_db"] - q["snr_db"]) * (a - np.log10(target)) / (a - b))
    return None


def deep_main(args, dtl, device):
    """--deep: both arithmetics over the same streams; few blocks where the curve is high, >= 20 000 per point where it
    is low; the dB distance between the arithmetics at BER 1e-3 and 1e-4 through the whole chain."""
    lo, hi, st = (float(v) for v in args.snr.split(":"))
    doc = dict(workload="mixed-MCS (BPSK/QPSK/8PSK/16QAM cycling) AWGN stream, %d-point FFT, frame_length %d, LDPC(300,152); "
                        "product GPU transmitter + device AWGN + product receiver" % (args.fft, args.frame_length),
               snr_definition="noise variance per time-domain sample = fft_len / 10^(snr/10)",
               ber_definition="(bits of blocks not delivered + wrong bits in delivered blocks) / bits sent (tools/ber.py:131-133)",
               sync_threshold=args.threshold, frame_length=args.frame_length, seed=args.seed, gap=args.gap,
               use_sync_correct=(args.gap == 0), curves={})
    for arith in ("gr_exact", "fast"):
        cfg = make_config(dtl, args.fft, args.frame_length, args.threshold, sync_correct=(args.gap == 0), arith=arith)
        pts = []
        snr = lo
        while snr < hi - 1e-9:
            deep = snr >= args.deep_from
            fps, cyc = (args.deep_frames_per_segment, args.deep_cycles) if deep else (32, 2)
            p = deep_point(dtl, cfg, snr, fps, cyc, args.seed + int(round(snr * 10)), device, gap=args.gap)
            pts.append(p)
            print("%-8s Es/N0 %5.1f dB  BER %.3e  BLER %.3e  blocks %d  frames %d  (16-QAM BLER %.3e)" % (
                arith, snr, p["ber"], p["bler"], p["blocks_sent"], p["frames_decoded"], p["per_constellation"]["qam16"]["bler"]),
                file=sys.stderr, flush=True)
            snr += st
        doc["curves"][arith] = pts
    gaps = {}
    for target in (1e-3, 1e-4):
        for sub in (None, "qam16", "psk8", "qpsk", "bpsk"):
            a = crossing_db(doc["curves"]["gr_exact"], "ber", target, sub)
            b = crossing_db(doc["curves"]["fast"], "ber", target, sub)
            gaps["%s@%g" % (sub or "all", target)] = dict(gr_exact_db=a, fast_db=b, gap_db=None if a is None or b is None else round(b - a, 3))
    doc["gap_fast_minus_gr_exact"] = gaps
    text = json.dumps(doc, indent=1)
    if args.out:
        open(args.out, "w").write(text + "\n")
    print(json.dumps(gaps, indent=1))


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
    ap.add_argument("--deep", action="store_true", help="GPU transmitter, both arithmetics, many blocks where the curve is low")
    ap.add_argument("--deep-from", type=float, default=22.0, help="Es/N0 from which a point gets the large stream")
    ap.add_argument("--deep-frames-per-segment", type=int, default=320)
    ap.add_argument("--deep-cycles", type=int, default=16)
    args = ap.parse_args()
    import torch
    import gr_dtl_amd as dtl
    if not torch.cuda.is_available():
        raise SystemExit("ber_sweep.py needs a HIP device: the product path has no CPU fallback")
    device = torch.device("cuda", 0)
    if args.deep:
        return deep_main(args, dtl, device)
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
