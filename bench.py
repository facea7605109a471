#!/usr/bin/env python3
"""bench.py -- adaptive-OFDM RX throughput (sync > FFT > eq > demap > LDPC) on N MI355X GPUs.

One "step" = one pass of the whole receive path (gr_dtl_amd.ofdm_receiver.work, i.e. dtl_rx_process
of the C ABI) over one batch of synthetic AWGN IQ frames that is already resident in HBM.
Workload (config.workload): BASELINE.json configs[2] -- 256-subcarrier OFDM, 16-QAM, LDPC(300,152),
64k-frame batch on one GPU; with --gpus N every rank owns its own 64k-frame shard (configs[3]:
512k frames over 8 GPUs) -> weak scaling, no data-path collective (frames are independent).

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel (LDPC BP) with SURVEY 8(d)'s
algorithmic bytes per sample; `cpu_baseline` times the CPU oracle (a port of the reference chain, see
oracle/) on a bounded sample of the same stream.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: fft_len, frame_length, constellation, Es/N0 dB, sync_threshold, frames, algorithmic B/sample
    "c2": dict(fft_len=64, frame_length=20, cnst=2, esn0_db=12.0, sync_threshold=0.95, frames=4096, bytes_per_sample=8.06,
               desc="64-subcarrier OFDM, QPSK, LDPC n=300 k=152, 4k-frame batch"),
    "c3": dict(fft_len=256, frame_length=20, cnst=4, esn0_db=23.0, sync_threshold=0.98, frames=65536, bytes_per_sample=8.13,
               desc="256-subcarrier OFDM, 16-QAM, LDPC(300,152), 64k-frame batch"),
}
ES = {1: 1.0, 2: 0.25, 3: 1.0, 4: 1.0}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def build_config(dtl, w):
    if w["fft_len"] == 64:
        cfg = dtl.ofdm_adaptive_rx_config(frame_length=w["frame_length"])
    else:
        cfg = dtl.make_scaled_rx_config(w["fft_len"], frame_length=w["frame_length"])
    cfg.sync_threshold = w["sync_threshold"]
    cfg.fec_codes = (("fec_1", "n_0300_k_0152_gap_03.alist"),)
    C = dtl.constellation_type_t
    cfg.mcs = ((-100000, (C.BPSK, "fec_1")), (11, (C.QPSK, "fec_1")), (16, (C.PSK8, "fec_1")), (21, (C.QAM16, "fec_1")))
    cfg.initial_mcs_id = 1
    return cfg


def synth_stream(dtl, cfg, w, n_frames, device, seed):
    """Synthetic AWGN IQ frames of the named FFT/codeword sizes: independent random payload for every frame,
    framed / LDPC-encoded / OFDM-modulated on the GPU by the product's own transmitter
    (gr_dtl_amd.ofdm_transmitter.work_device), 100 leading zeros, one frame of trailing zeros, plus
    independent complex AWGN.  -> (iq, frames, user bytes sent, user tensor)"""
    import torch
    from gr_dtl_amd.ofdm_transmitter import ofdm_transmitter
    tx = ofdm_transmitter(cfg)
    bps = int(w["cnst"])
    frame_bytes = tx.frame_capacity * bps // 8
    g = torch.Generator(device=device)
    g.manual_seed(0xD71 + seed)
    # user bytes that fill n_frames frames: whole transport blocks of the LDPC(300,152) code (tb_encoder split)
    n_code, k_code = 300, 152
    ncws = 1 + frame_bytes * 8 // n_code if frame_bytes * 8 > n_code else 1
    tb_payload = ncws * k_code // 8 - 4
    tb_coded_bytes = (ncws * (n_code - k_code) + (tb_payload + 4) * 8) / 8.0
    n_user = int(n_frames * frame_bytes / tb_coded_bytes) * tb_payload
    user = torch.randint(0, 256, (n_user,), dtype=torch.uint8, device=device, generator=g)
    iq, hdrs, _ = tx.work_device(user, w["cnst"], "fec_1")
    lead = torch.zeros(100, dtype=torch.complex64, device=device)             # qa_ofdm_adaptive_txrx.py:82-83
    tail = torch.zeros(tx.frame_samples, dtype=torch.complex64, device=device)
    x = torch.cat([lead, iq.reshape(-1), tail])
    del iq
    nv = w["fft_len"] * ES[w["cnst"]] / 10 ** (w["esn0_db"] / 10)
    g.manual_seed(1234 + seed)
    noise = torch.randn((x.numel(), 2), generator=g, device=device, dtype=torch.float32) * float(np.sqrt(nv / 2))
    x += torch.view_as_complex(noise)
    del noise
    return x, len(hdrs), n_user, user


def cpu_baseline(cfg, w, x_host, n_frames_sample, frame_samples, cores):
    """The CPU oracle (C restatement of the reference chain) on a bounded sample of the same stream, spread over
    `cores` host threads: the sample is cut into contiguous runs of whole frames, one run per thread (ctypes
    releases the GIL during the C call; every thread has its own decoder object)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    oracle.build()
    alist = os.path.join(ROOT, "gr_dtl_amd", "codes", "n_0300_k_0152_gap_03.alist")

    def make():
        ocfg = oracle.Config(fft_len=cfg.fft_len, cp_len=cfg.cp_len, frame_length=cfg.frame_length, has_fec=True,
                             sync_threshold=cfg.sync_threshold, occupied_carriers=cfg.occupied_carriers,
                             pilot_carriers=cfg.pilot_carriers, pilot_symbols=cfg.pilot_symbols,
                             sync_word1=cfg.sync_word1, sync_word2=cfg.sync_word2)
        return ocfg, oracle.Ldpc(alist)
    cores = max(1, min(cores, n_frames_sample))
    pieces = []
    for c in range(cores):
        lo, hi = n_frames_sample * c // cores, n_frames_sample * (c + 1) // cores
        # 100 samples of lead-in (zeros for the first run, the end of the previous frame otherwise) and half a
        # frame behind the last one, as the whole stream has
        pieces.append(np.ascontiguousarray(x_host[lo * frame_samples:100 + hi * frame_samples + frame_samples // 2]))
    ctx = [make() for _ in range(cores)]
    oracle.rx_stream(ctx[0][0], ctx[0][1], pieces[0][:100 + 2 * frame_samples])     # warm-up (page-in, tables)

    def run(i):
        return oracle.rx_stream(ctx[i][0], ctx[i][1], pieces[i])["n_frames"]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as pool:
        decoded = sum(pool.map(run, range(cores)))
    dt = time.perf_counter() - t0
    n_samples = sum(p.size for p in pieces)
    return dict(value=round(n_samples / dt / 1e6, 4), unit="Msamples/s", cores=cores, kind="port",
                sample="first %d frames (%d samples) of the same stream in %d runs of whole frames, one host thread "
                       "each; oracle/ C restatement of the reference chain; %.1f s wall, %.1f core-seconds; "
                       "%d frames decoded" % (n_frames_sample, n_samples, cores, dt, dt * cores, decoded))


def launch_ranks(n, argv):
    """One process per GPU: `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py argv`, rendezvous on
    127.0.0.1 at a free port.  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    print("bench.py: starting %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU (default: the workload's batch)")
    ap.add_argument("--in-flight", type=int, default=1, choices=(1, 2),
                    help="receivers decoding alternate steps concurrently on their own HIP streams / host threads "
                         "(2: the host phases and the sync kernel of one batch overlap the decoder kernels of the other)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0)
    ap.add_argument("--cpu-cores", type=int, default=0, help="host threads of the CPU baseline (default: all, at most 16)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves.  This parent never touches
        # HIP (no torch import, no device query): the ranks are children of torch.distributed.run, their stdout
        # (rank 0's ONE JSON line) is ours.
        return launch_ranks(args.gpus, sys.argv[1:])
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s; launch with --nproc-per-node equal to --gpus "
                         "(or without a launcher: bench.py starts the ranks itself)"
                         % (args.gpus, os.environ.get("WORLD_SIZE", "unset")))

    # stdout carries exactly ONE JSON line: native libraries (RCCL prints a version banner on fd 1) and any
    # stray print are sent to stderr; the result goes to the saved descriptor at the end
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import gr_dtl_amd as dtl

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    # rehearsal switches for a one-GPU box (the real multi-GPU run uses neither): every rank on device 0 with the
    # gloo backend exercises the multi-process path; DTL_BENCH_FORCE_DIST initialises RCCL with a single rank
    rehearsal = os.environ.get("DTL_BENCH_REHEARSAL_GLOO") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    force_dist = os.environ.get("DTL_BENCH_FORCE_DIST") == "1"
    if world > 1 or (force_dist and "RANK" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)  # RCCL over xGMI

    w = WORKLOADS[args.workload]
    n_frames_req = args.frames or w["frames"]
    cfg = build_config(dtl, w)
    rx = dtl.ofdm_receiver(cfg)
    x, n_frames, n_user, _ = synth_stream(dtl, cfg, w, n_frames_req, device, seed=rank)
    n_samples = x.numel()
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    res = None
    ldpc_ms, payload_ms, stage = [], [], np.zeros(8)
    if args.in_flight == 1:
        for _ in range(args.warmup):
            res = rx.work(x, copy_outputs=False)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            # results stay where the C ABI leaves them (payload bytes in HBM, per-frame / per-block records in the
            # handle's host memory); the numpy copies of the records are made on first read (after the timed region)
            res = None
            res = rx.work(x, copy_outputs=False)
            ldpc_ms.append(res.stage_ms[6])
            payload_ms.append(res.stage_ms[7])
            stage += np.array(res.stage_ms)
        barrier()
        elapsed = time.perf_counter() - t0
    else:
        import threading
        rxs = [rx, dtl.ofdm_receiver(cfg)]
        streams = [torch.cuda.Stream(device=device) for _ in rxs]
        last = [None, None]
        share = [args.steps - args.steps // 2, args.steps // 2]        # exactly K steps in total

        def run(i, count):
            torch.cuda.set_device(device)
            with torch.cuda.stream(streams[i]):
                for _ in range(count):
                    r = last[i] = None
                    r = rxs[i].work(x, copy_outputs=False)
                    last[i] = r
                    ldpc_ms.append(r.stage_ms[6])
                    payload_ms.append(r.stage_ms[7])

        def both(counts):
            ths = [threading.Thread(target=run, args=(i, counts[i])) for i in range(2)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        both([max(1, args.warmup - args.warmup // 2), max(1, args.warmup // 2)])
        barrier()
        t0 = time.perf_counter()
        both(share)
        barrier()
        elapsed = time.perf_counter() - t0
        res = last[0]
        stage = np.array(res.stage_ms) * args.steps
    from gr_dtl_amd.sharding import combine
    # max-over-ranks time, summed counters (two tiny all-reduces; no data-path collective)
    elapsed, tot = combine(elapsed, [n_samples, res.n_payload_bytes, res.n_payload_frames, res.n_codewords],
                           dist, None if rehearsal else device)
    total_samples = float(tot[0])
    ms_per_step = elapsed / args.steps * 1e3
    value = total_samples * args.steps / elapsed / 1e6

    out = None
    if rank == 0:
        extra = rx.work(x, want=("cw",))                      # untimed: per-codeword iteration counts
        cw_it = float(extra.cw_nit.float().mean().item()) if extra.cw_nit is not None else 0.0
        cw_max = float((extra.cw_nit == 15).float().mean().item()) if extra.cw_nit is not None else 0.0
        ok_tbs = int((res.tbs["crc_ok"] == 1).sum())
        avg_it = float(res.tbs["avg_it"].mean()) if len(res.tbs) else 0.0
        t_ldpc = float(np.mean(ldpc_ms)) if ldpc_ms else 0.0
        alg_bytes = w["bytes_per_sample"] * n_samples            # bytes/sample x samples per launch
        achieved = alg_bytes / (t_ldpc * 1e-3) / 1e9 if t_ldpc > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("ldpc_bp_kernel_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Msamples/s thru OFDM RX (sync>FFT>eq>demap>LDPC)", "value": round(value, 3),
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (sync/FFT/eq/demap) + f64 (LDPC BP)", "data": "synthetic (unique random payload per frame, product GPU TX + AWGN)",
            "config": {"workload": w["desc"], "fft_len": w["fft_len"], "frames_per_gpu": n_frames,
                       "samples_per_gpu": n_samples, "constellation": int(w["cnst"]), "esn0_db": w["esn0_db"],
                       "sync_threshold": w["sync_threshold"], "ldpc": "n=300 k=152, <=15 it, f64 LR-domain SPA",
                       "parallelism": "frame-batch shard per GPU (dp%d), no data-path collective" % world,
                       "batches_in_flight_per_gpu": args.in_flight},
            "roofline": {"bound": "hbm", "kernel": "ldpc_bp_np_kernel", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                         "traffic": traffic, "kernel_ms": round(t_ldpc, 3),
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "note": "kernel is FP64-VALU/LDS bound (4 f64 divisions per Tanner edge per iteration), "
                                 "not HBM bound; see DESIGN.md"},
            "stats": {"frames_decoded": int(res.n_payload_frames), "tbs": int(len(res.tbs)), "tbs_crc_ok": ok_tbs,
                      "avg_bp_iterations_per_tb": round(avg_it, 3), "bp_iterations_per_codeword": round(cw_it, 3),
                      "codewords_at_max_iterations": round(cw_max, 4), "codewords": int(res.n_codewords),
                      "payload_bytes": int(res.n_payload_bytes), "user_bytes_sent": int(n_user),
                      "payload_kernel_ms": round(float(np.mean(payload_ms)), 3),
                      "host_stage_ms": [round(float(s) / args.steps, 3) for s in stage[:6]],
                      "whole_chain_hbm_GBps": round(w["bytes_per_sample"] * value / 1e3, 2)},
        }
        if not args.no_cpu_baseline:
            cores = args.cpu_cores or min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
            nfs = args.cpu_frames or min(n_frames, 1024 * cores)          # ~1.4 s of CPU work per 1024 C3 frames
            nfs = min(nfs, n_frames)
            ns = 100 + nfs * rx.frame_samples + rx.frame_samples // 2
            out["cpu_baseline"] = cpu_baseline(cfg, w, x[:ns].cpu().numpy(), nfs, rx.frame_samples, cores)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    rc = main()
    sys.exit(rc if isinstance(rc, int) else 0)
