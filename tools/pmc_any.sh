#!/bin/bash
# Run on the GPU box: one rocprofv3 counter pass (<= 8 SQ counters) of the default bench; per-kernel per-launch averages.
# usage: pmc_any.sh <tag> COUNTER...
set -euo pipefail
TAG="$1"; shift
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$R/gpurun_out/pmc_${TAG}"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --output-format csv -d "$OUT/p1" -o c3 -- python3 "$R/bench.py" --steps 2 --warmup 1 --headline-only > "$OUT/p1.json" 2> "$OUT/p1.err"
python3 "$R/tools/pmc_sq.py" "$OUT" > "$OUT/summary.txt"
find "$OUT" -name "*.csv" -size +20M -delete
cat "$OUT/summary.txt"
