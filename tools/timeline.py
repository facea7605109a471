#!/usr/bin/env python3
"""Device timeline of one bench step from a rocprofv3 --kernel-trace CSV: for every kernel of the last step its
start relative to the step's first kernel, duration, and the idle gap on the device before it (time during which
NO kernel of the step was running).  usage: timeline.py <kernel_trace.csv>"""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (re.findall(r"(\w+_kernel|__amd_rocclr_\w+)", r["Kernel_Name"]) or [r["Kernel_Name"][:40]])[0]))
rows.sort()
# steps start with the first sync_edges_kernel after an ldpc kernel
step_starts = []
seen_ldpc = True
for i, r in enumerate(rows):
    if "ldpc_bp" in r[2]:
        seen_ldpc = True
    elif "sync_edges" in r[2] and seen_ldpc:
        step_starts.append(i)
        seen_ldpc = False
if len(step_starts) < 2:
    sys.exit("no complete step found")
a, b = step_starts[-2], step_starts[-1]
step = rows[a:b]
t0 = step[0][0]
busy_until = t0
idle = 0
print(f"{'kernel':42s} {'start ms':>9s} {'dur ms':>8s} {'idle before':>11s}")
for s, e, n in step:
    gap = max(0, s - busy_until)
    idle += gap
    print(f"{n:42s} {(s - t0) / 1e6:9.3f} {(e - s) / 1e6:8.3f} {gap / 1e6:11.3f}")
    busy_until = max(busy_until, e)
print(f"step span {(busy_until - t0) / 1e6:.3f} ms, device idle inside it {idle / 1e6:.3f} ms, kernel time sum {sum(e - s for s, e, _ in step) / 1e6:.3f} ms")
