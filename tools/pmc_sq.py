#!/usr/bin/env python3
"""Condense the counter_collection CSVs of tools/pmc_sq.sh: per kernel, per-launch average of every SQ counter and a
few ratios (VALU instructions per wave, share of wave cycles parked / issue-stalled / issuing, LDS conflict share)."""
import collections
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for p in ("p1", "p2"):
    for path in glob.glob(os.path.join(root, p, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = (re.findall(r"(\w+_kernel)", r["Kernel_Name"]) or [r["Kernel_Name"][:32]])[0]
                a = acc[name][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
want = ["sync_edges_kernel", "header_kernel", "frame_eq_kernel", "demap_kernel", "ldpc_bp_np_kernel", "tb_check_kernel"]
for k in want:
    if k not in acc:
        continue
    v = {c: s / max(n, 1) for c, (s, n) in acc[k].items()}
    print(f"== {k}")
    for c in sorted(v):
        print(f"   {c:24s} {v[c]:16.0f}")
    wc = v.get("SQ_WAVE_CYCLES", 0)
    if wc:
        print(f"   share of wave cycles: parked {v.get('SQ_WAIT_ANY', 0) / wc:.3f}  issue-stalled {v.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}  "
              f"issuing {v.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f} (VALU {v.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f}, LDS {v.get('SQ_ACTIVE_INST_LDS', 0) / wc:.3f})")
    if v.get("SQ_WAVES"):
        print(f"   per wave: VALU {v.get('SQ_INSTS_VALU', 0) / v['SQ_WAVES']:.0f}  SALU {v.get('SQ_INSTS_SALU', 0) / v['SQ_WAVES']:.0f}  "
              f"LDS {v.get('SQ_INSTS_LDS', 0) / v['SQ_WAVES']:.0f}  VMEM rd {v.get('SQ_INSTS_VMEM_RD', 0) / v['SQ_WAVES']:.0f} wr {v.get('SQ_INSTS_VMEM_WR', 0) / v['SQ_WAVES']:.0f}")
    if v.get("SQ_LDS_IDX_ACTIVE"):
        print(f"   LDS bank-conflict cycles / LDS active cycles: {v.get('SQ_LDS_BANK_CONFLICT', 0) / v['SQ_LDS_IDX_ACTIVE']:.3f}")
