#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel statistics + HBM traffic counters of the default bench.
# (the profiled command is `bench.py --headline-only`: the headline step only, so that the per-kernel
# averages are those of the bench line's kernels and not mixed with the secondary measurements of `extra`)
# Three separate passes (kernel trace; FETCH_SIZE; WRITE_SIZE) as MI355X_MICROARCH.md prescribes.
# Results land in gpurun_out/prof_<tag>/ ; tools/pmc_traffic.py condenses them into profiles/.
set -euo pipefail
TAG="${1:-r01}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$R/gpurun_out/prof_${TAG}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o c3 -- python3 "$R/bench.py" --steps 5 --warmup 2 --headline-only > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
echo "kernel-trace pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o c3 -- python3 "$R/bench.py" --steps 2 --warmup 1 --headline-only > "$OUT/fetch.json" 2> "$OUT/fetch.err"
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o c3 -- python3 "$R/bench.py" --steps 2 --warmup 1 --headline-only > "$OUT/write.json" 2> "$OUT/write.err"
echo "WRITE_SIZE pass done"
python3 "$R/bench.py" --steps 10 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 "$R/tools/pmc_traffic.py" "$OUT" "$TAG" c3
# the device timeline of ONE call: a trace of plain calls (with two batches in flight on the handle's lanes -- the bench's
# default -- the kernels of two steps interleave and no step can be cut out of the trace)
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_plain" -o c3 -- python3 "$R/bench.py" --steps 5 --warmup 2 --headline-only --no-pipeline > "$OUT/trace_plain.json" 2> "$OUT/trace_plain.err"
KT="$(find "$OUT/trace_plain" -name '*kernel_trace.csv' | head -1)"
if [[ -n "$KT" ]]; then python3 "$R/tools/timeline.py" "$KT" > "$OUT/summary/c3_timeline.txt" || true; fi
find "$OUT" -name "*.csv" -size +20M -delete   # raw per-dispatch counter dumps can be large
ls -la "$OUT" "$OUT"/*
