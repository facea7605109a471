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
# A "step" here runs from one call's first sync kernel to the next call's.  The last such interval of a bench run ends at
# whatever follows the timed loop (the max-over-ranks tensors of bench.combine, the statistics pass) -- rounds 1-2 printed
# that one and showed four small copies and > 1 ms of "idle" that belong to the bench's epilogue, not to a step.  Take the
# interval of median length among those that are followed by another step of the same loop.
spans = [(rows[step_starts[i + 1]][0] - rows[step_starts[i]][0], i) for i in range(len(step_starts) - 1)]
inner = spans[:-1] if len(spans) > 2 else spans
inner.sort()
pick = inner[len(inner) // 2][1]
a, b = step_starts[pick], step_starts[pick + 1]
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
tail_gap = max(0, rows[b][0] - busy_until)     # from this step's last kernel to the next call's first
print(f"step span {(busy_until - t0) / 1e6:.3f} ms, device idle inside it {idle / 1e6:.3f} ms, kernel time sum {sum(e - s for s, e, _ in step) / 1e6:.3f} ms; "
      f"then {tail_gap / 1e6:.3f} ms until the next call's first kernel (host: block records, caller, next call's launches) "
      f"= {(idle + tail_gap) / 1e6:.3f} ms of device idle per {(rows[b][0] - t0) / 1e6:.3f} ms period")
