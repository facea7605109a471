#!/bin/bash
# Run on the GPU box: rocprofv3 kernel statistics of one secondary bench workload (c2, c5) -> gpurun_out/prof_<tag>/<w>_*
# (the headline workload has its own script with the traffic passes: tools/profile_c3.sh)
set -euo pipefail
W="${1:?workload}"; TAG="${2:-r01}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$R/gpurun_out/prof_${TAG}"
mkdir -p "$OUT/summary"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$W" -o "$W" -- python3 "$R/bench.py" --workload "$W" --steps 10 --warmup 3 --headline-only > "$OUT/summary/${W}_bench_under_rocprof.json" 2> "$OUT/stats_$W.err"
KS="$(find "$OUT/stats_$W" -name '*kernel_stats.csv' | head -1)"
cp "$KS" "$OUT/summary/${W}_kernel_stats.csv"
python3 "$R/bench.py" --workload "$W" --steps 20 --warmup 3 --headline-only > "$OUT/summary/${W}_bench.json" 2> "$OUT/${W}_bench.err"
find "$OUT/stats_$W" -name "*.csv" -size +20M -delete
head -8 "$OUT/summary/${W}_kernel_stats.csv"
