#!/bin/bash
# Timing experiments on the GPU box: libraries built with -DDTL_ABLATE=k (gr_dtl_amd/csrc/_obj/ablate_k.so; results are
# wrong by construction, only the kernel times mean something).  Prints the rocprofv3 average of the phase B kernels.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
cp "$R/gr_dtl_amd/libdtl_hip.so" /tmp/keep.so
for A in "$@"; do
  cp "$R/gr_dtl_amd/csrc/_obj/ablate_$A.so" "$R/gr_dtl_amd/libdtl_hip.so"
  rm -rf /tmp/abl_$A
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$A -o a -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/abl_$A.err
  echo "== ablate $A"
  python3 - "$A" <<'PY'
import csv, glob, sys
for p in glob.glob('/tmp/abl_%s/**/*kernel_stats.csv' % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(p)):
        if any(k in r["Name"] for k in ("frame_eq", "demap_kernel", "header_kernel", "ldpc_bp_np", "sync_edges", "tb_check")):
            print("   %-28s calls %4s avg %9.1f us" % (r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
cp /tmp/keep.so "$R/gr_dtl_amd/libdtl_hip.so"
