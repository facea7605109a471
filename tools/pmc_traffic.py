#!/usr/bin/env python3
"""Condense a tools/profile_c3.sh run: per-kernel duration statistics (rocprofv3 --kernel-trace --stats) and
HBM traffic per launch from the FETCH_SIZE / WRITE_SIZE passes (units and the gfx950 x2 FETCH_SIZE correction as
in MI355X_MICROARCH.md) -> <out_dir>/summary/ (copy it to profiles/<tag>/; traffic_<workload>.json goes to profiles/,
where bench.py reads it)."""
import csv
import glob
import json
import os
import shutil
import sys

KERNELS = {"sync_edges": "sync_edges_kernel", "trigger_angle": "trigger_angle_kernel", "header_kernel": "header_kernel",
           "frame_eq": "frame_eq_kernel", "demap": "demap_kernel", "expand_cw": "expand_cw_kernel", "ldpc_bp": "ldpc_bp_np_kernel",
           "tb_check": "tb_check_kernel"}


def per_kernel_counter(root, counter):
    """average counter value per launch for each of our kernels"""
    acc = {}
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"]
                for key, pat in KERNELS.items():
                    if pat in name:
                        a = acc.setdefault(key, [0.0, 0])
                        a[0] += float(row["Counter_Value"])
                        a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}


def main():
    out_dir, tag, workload = sys.argv[1], sys.argv[2], sys.argv[3]
    dst = os.path.join(out_dir, "summary")
    os.makedirs(dst, exist_ok=True)
    for path in glob.glob(os.path.join(out_dir, "stats", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(path, os.path.join(dst, "%s_kernel_stats.csv" % workload))
    for name in ("bench.json", "bench_under_rocprof.json"):
        p = os.path.join(out_dir, name)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(dst, "%s_%s" % (workload, name)))
    fetch = per_kernel_counter(os.path.join(out_dir, "fetch"), "FETCH_SIZE")
    write = per_kernel_counter(os.path.join(out_dir, "write"), "WRITE_SIZE")
    kernels = {}
    steps = fetch.get("ldpc_bp", (0, 1))[1]      # the LDPC kernel is launched once per step
    for k in KERNELS:
        if k in fetch or k in write:
            fb = fetch.get(k, (0.0, 0))[0] * 1024 * 2   # FETCH_SIZE counts KiB; gfx950 reports half (MI355X_MICROARCH.md)
            wb = write.get(k, (0.0, 0))[0] * 1024
            per_step = fetch.get(k, write.get(k))[1] / max(steps, 1)   # phases A/B and the sync stage run in 4 chunks
            kernels[k] = {"fetch_bytes_corrected": fb, "write_bytes": wb, "hbm_bytes": fb + wb,
                          "launches_per_step": per_step, "hbm_bytes_per_step": (fb + wb) * per_step}
    doc = {"workload": workload,
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x1024 B x2 (gfx950 half-count "
                   "correction of MI355X_MICROARCH.md), WRITE_SIZE x1024 B; per launch averages",
           "kernels": kernels,
           "ldpc_bp_kernel_hbm_bytes_per_launch": kernels.get("ldpc_bp", {}).get("hbm_bytes"),
           # phase B = frame equaliser (+ soft demapper, its own kernel or fused), summed over its launches of one step
           "phase_b_hbm_bytes_per_step": (kernels.get("frame_eq", {}).get("hbm_bytes_per_step", 0.0) +
                                          kernels.get("demap", {}).get("hbm_bytes_per_step", 0.0)) or None}
    if kernels:
        json.dump(doc, open(os.path.join(dst, "traffic_%s.json" % workload), "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
