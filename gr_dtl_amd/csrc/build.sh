#!/bin/bash
# Build libdtl_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../libdtl_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
SRCS=("${HERE}"/*.hip "${HERE}"/*.cpp)
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-result)
mkdir -p "${HERE}/_obj"
OBJS=()
for s in "${SRCS[@]}"; do
  o="${HERE}/_obj/$(basename "$s").o"
  # headers: the kernels' own (csrc/*.h) and the ABI structs / status codes (include/*.h, *.hpp)
  if [[ ! -f "$o" || "$s" -nt "$o" || -n "$(find "${HERE}" "${HERE}/../../include" -maxdepth 1 \( -name '*.h' -o -name '*.hpp' \) -newer "$o" -print -quit)" ]]; then
    "${HIPCC}" "${FLAGS[@]/-shared/}" -x hip -c "$s" -o "$o" ${EXTRA_HIPCC_FLAGS:-}
  fi
  OBJS+=("$o")
done
"${HIPCC}" --offload-arch=gfx950 -fPIC -shared -o "${OUT}" "${OBJS[@]}"
echo "built ${OUT}"
