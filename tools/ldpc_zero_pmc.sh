#!/bin/bash
# Run on the GPU box: SQ counters of tools/ldpc_probe.py (counters only, no tracing), three passes.
set -uo pipefail
TAG="${1:-zero}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$R/gpurun_out/sqz_${TAG}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS \
  --output-format csv -d "$OUT/p1" -o z -- python3 "$R/tools/ldpc_probe.py" > "$OUT/p1.txt" 2> "$OUT/p1.err"
echo "pass 1 rc $?"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
  --output-format csv -d "$OUT/p2" -o z -- python3 "$R/tools/ldpc_probe.py" > "$OUT/p2.txt" 2> "$OUT/p2.err"
echo "pass 2 rc $?"
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SENDMSG SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT \
  --output-format csv -d "$OUT/p3" -o z -- python3 "$R/tools/ldpc_probe.py" > "$OUT/p3.txt" 2> "$OUT/p3.err"
echo "pass 3 rc $?"
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES \
  --output-format csv -d "$OUT/p4" -o z -- python3 "$R/tools/ldpc_probe.py" > "$OUT/p4.txt" 2> "$OUT/p4.err"
echo "pass 4 rc $?"
python3 - "$OUT" <<'PY'
import collections, csv, glob, os, sys
root = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            if "ldpc_bp_np" not in r["Kernel_Name"]:
                continue
            a = acc[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for c in sorted(acc):
    print("%-26s %16.0f  (%d launches)" % (c, acc[c][0] / acc[c][1], acc[c][1]))
PY
cat "$OUT"/p1.txt; tail -2 "$OUT"/p3.err "$OUT"/p4.err
find "$OUT" -name "*.csv" -size +20M -delete
