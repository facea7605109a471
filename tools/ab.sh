#!/bin/bash
# A/B of two builds of libdtl_hip.so on the GPU box: gr_dtl_amd/csrc/_obj/variant_{old,new}.so, alternating runs of
# the default bench; leaves the "new" library in place.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$R"
for V in old new old new; do
  cp gr_dtl_amd/csrc/_obj/variant_$V.so gr_dtl_amd/libdtl_hip.so && python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/v_$V.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/v_$V.json')); print('$V', d['ms_per_step'], d['roofline']['kernel_ms'], d['stats'].get('payload_kernels_ms'), d['stats']['tbs_crc_ok'], d['stats']['host_stage_ms'])"
done
