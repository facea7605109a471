#!/bin/bash
# Timing experiments for the sync kernel: libraries built with -DDTL_ABLATE=k (gr_dtl_amd/csrc/_obj/ablate_k.so), "base" =
# the library as built; DTL_SYNC_NO_PRETEST=1 as variant "nopre".  Prints the rocprofv3 average of sync_edges_kernel.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
cp "$R/gr_dtl_amd/libdtl_hip.so" /tmp/keep.so
for A in "$@"; do
  unset DTL_SYNC_NO_PRETEST
  if [[ "$A" == "nopre" ]]; then export DTL_SYNC_NO_PRETEST=1; cp /tmp/keep.so "$R/gr_dtl_amd/libdtl_hip.so";
  elif [[ "$A" == "base" ]]; then cp /tmp/keep.so "$R/gr_dtl_amd/libdtl_hip.so";
  else cp "$R/gr_dtl_amd/csrc/_obj/ablate_$A.so" "$R/gr_dtl_amd/libdtl_hip.so"; fi
  rm -rf /tmp/abl_$A
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$A -o a -- python3 "$R/tools/bench_sync.py" > /tmp/abl_$A.out 2> /tmp/abl_$A.err
  echo "== $A: $(tail -1 /tmp/abl_$A.out)"
  python3 - "$A" <<'PY'
import csv, glob, sys
for p in glob.glob('/tmp/abl_%s/**/*kernel_stats.csv' % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(p)):
        if "sync_edges" in r["Name"]:
            print("   %-60s calls %4s avg %9.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
cp /tmp/keep.so "$R/gr_dtl_amd/libdtl_hip.so"
