#!/bin/bash
# Build timing-experiment variants of the library: gr_dtl_amd/csrc/_obj/ablate_<k>.so = libdtl_hip.so with rx_kernels.hip
# compiled -DDTL_ABLATE=<k> (results are wrong by construction; only kernel times mean something).
#   1 / 2 / 3   frame equaliser without derotation / butterflies / equaliser (tools/ablate.sh)
#   11 / 12     sync kernel without its window test / without strip totals either (tools/ablate_sync.sh)
#   31 - 34     demapper without exponentials / logarithms / stores / symbol loads (tools/ablate.sh)
# Run here (hipcc cross-compiles), then tools/ablate.sh or tools/ablate_sync.sh on the GPU box.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/../gr_dtl_amd/csrc" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
bash "$HERE/build.sh" > /dev/null
for k in "$@"; do
  "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -DDTL_ABLATE="$k" -x hip -c "$HERE/rx_kernels.hip" -o "/tmp/rx_kernels_ablate_$k.o"
  OBJS=()
  for o in "$HERE"/_obj/*.o; do [[ "$(basename "$o")" == "rx_kernels.hip.o" ]] || OBJS+=("$o"); done
  "$HIPCC" --offload-arch=gfx950 -fPIC -shared -o "$HERE/_obj/ablate_$k.so" "/tmp/rx_kernels_ablate_$k.o" "${OBJS[@]}"
  echo "built $HERE/_obj/ablate_$k.so"
done
