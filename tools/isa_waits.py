#!/usr/bin/env python3
"""Where a kernel can wait for memory: per basic block of a `hipcc -S --cuda-device-only` listing, the global / scratch
/ flat operations, the `s_waitcnt vmcnt(..)` lines and the barriers, with the block's loop depth.  A load followed
by `vmcnt(0)` in the same block inside a loop is an exposed memory latency per pass; `scratch_` lines are spills;
`flat_load` means a pointer that may be LDS or global (its wait drains both queues).

usage: isa_waits.py <listing.s> <kernel name substring> [min loop depth]
       hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -S --cuda-device-only -Iinclude -Igr_dtl_amd/csrc \\
             -o /tmp/rx.s gr_dtl_amd/csrc/rx_kernels.hip"""
import itertools
import re
import sys

txt = open(sys.argv[1]).read()
names = [m.group(1) for m in re.finditer(r"^(\S*%s\S*):" % re.escape(sys.argv[2]), txt, re.M)]
if not names:
    sys.exit("no such kernel in the listing")
min_depth = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for name in names:
    start = txt.index(name + ":")
    end = txt.index(".Lfunc_end", start)
    cur, depth, out = "entry", {"entry": 0}, []
    for line in txt[start:end].split("\n"):
        ls = line.strip()
        m = re.match(r"^\.LBB\d+_(\d+):", ls)
        if m:
            cur = m.group(1)
            d = re.search(r"Depth=(\d+)", ls)
            depth[cur] = int(d.group(1)) if d else 0
            continue
        op = ls.split(" ")[0] if ls else ""
        if op.startswith(("global_", "scratch_", "flat_", "buffer_")) or (op == "s_waitcnt" and "vmcnt" in ls) or op == "s_barrier":
            out.append((cur, ls.split(";")[0].strip()))
    print("==", name[:100])
    for k, g in itertools.groupby(out, key=lambda x: x[0]):
        if depth.get(k, 0) >= min_depth:
            print("  BB%-4s depth %d  %s" % (k, depth.get(k, 0), " | ".join(x[1][:46] for x in g)[:320]))
