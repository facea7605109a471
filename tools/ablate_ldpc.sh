#!/bin/bash
# Timing experiments on the decoder's per-codeword fixed cost.  Here (no GPU): `tools/ablate_ldpc.sh build 21 22 23`
# builds gr_dtl_amd/csrc/_obj/ablate_<k>.so = the library with ldpc_bp.hip compiled -DDTL_ABLATE=<k> (results are
# wrong by construction).  On the GPU box: `tools/ablate_ldpc.sh run 21 22 23` times tools/ldpc_zero_probe.py with the
# shipped library and with each variant, and puts the shipped library back.
set -euo pipefail
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
HERE="$R/gr_dtl_amd/csrc"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
MODE="$1"; shift
if [[ "$MODE" == build ]]; then
  bash "$HERE/build.sh" > /dev/null
  for k in "$@"; do
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -I"$R/include" -DDTL_ABLATE="$k" -x hip -c "$HERE/ldpc_bp.hip" -o "/tmp/ldpc_bp_ablate_$k.o"
    OBJS=()
    for o in "$HERE"/_obj/*.o; do [[ "$(basename "$o")" == "ldpc_bp.hip.o" ]] || OBJS+=("$o"); done
    "$HIPCC" --offload-arch=gfx950 -fPIC -shared -o "$HERE/_obj/ablate_$k.so" "/tmp/ldpc_bp_ablate_$k.o" "${OBJS[@]}"
    echo "built $HERE/_obj/ablate_$k.so"
  done
else
  cp "$R/gr_dtl_amd/libdtl_hip.so" /tmp/libdtl_hip_shipped.so
  echo "shipped: $(python3 "$R/tools/ldpc_zero_probe.py" 2>/dev/null)"
  for k in "$@"; do
    cp "$HERE/_obj/ablate_$k.so" "$R/gr_dtl_amd/libdtl_hip.so"
    echo "ablate $k: $(python3 "$R/tools/ldpc_zero_probe.py" 2>/dev/null)"
  done
  cp /tmp/libdtl_hip_shipped.so "$R/gr_dtl_amd/libdtl_hip.so"
  echo "shipped: $(python3 "$R/tools/ldpc_zero_probe.py" 2>/dev/null)"
fi
