#!/bin/bash
# Run on the GPU box: SQ counters of the default bench, two passes of <= 8 counters (counters only, no tracing).
# Output: gpurun_out/sq_<tag>/summary.txt (per kernel: per-launch averages) -- copy into profiles/ to keep.
set -euo pipefail
TAG="${1:-r01}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$R/gpurun_out/sq_${TAG}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS \
  --output-format csv -d "$OUT/p1" -o c3 -- python3 "$R/bench.py" --steps 2 --warmup 1 --headline-only > "$OUT/p1.json" 2> "$OUT/p1.err"
echo "pass 1 done"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
  --output-format csv -d "$OUT/p2" -o c3 -- python3 "$R/bench.py" --steps 2 --warmup 1 --headline-only > "$OUT/p2.json" 2> "$OUT/p2.err"
echo "pass 2 done"
python3 "$R/tools/pmc_sq.py" "$OUT" > "$OUT/summary.txt"
find "$OUT" -name "*.csv" -size +20M -delete
cat "$OUT/summary.txt"
