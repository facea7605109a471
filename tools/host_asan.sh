#!/bin/bash
# Host-side AddressSanitizer + UBSan run of the product library (device code is compiled as usual; the GPU pool has no
# sanitizer support, so this runs the tests that need no GPU: trigger fixing, plateau detector, feedback decision, code
# tables, encoder, TX framing, header formatter, ABI checks, the bench launcher) -- and the oracle under the same
# sanitizers for good measure.  Usage: tools/host_asan.sh   (from the repo root; ~5 min)
set -euo pipefail
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OBJ="$(mktemp -d)"
for s in "$R"/gr_dtl_amd/csrc/*.hip "$R"/gr_dtl_amd/csrc/*.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=address,undefined \
      -fno-gpu-sanitize -fno-omit-frame-pointer -x hip -c "$s" -o "$OBJ/$(basename "$s").o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan \
    -o "$OBJ/libdtl_hip_asan.so" "$OBJ"/*.o
gcc -g -O1 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -ffp-contract=off -fPIC \
    -std=gnu11 -shared -o "$OBJ/liboracle_asan.so" "$R"/oracle/*.c -lm -lpthread
RT="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)"
cd "$R"
echo "== product library, host sanitizers"
DTL_HIP_LIB="$OBJ/libdtl_hip_asan.so" LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 \
    python -m pytest tests -q -m "not gpu" -p no:cacheprovider --deselect tests/test_abi_exceptions_cpu.py
# (tests/test_abi_exceptions_cpu.py asks for allocations no machine has and expects std::bad_alloc / std::length_error to
# arrive at the boundary as a status: the sanitizer's allocator aborts on such a request instead of throwing)
echo "== oracle, sanitizers"
DTL_ORACLE_LIB="$OBJ/liboracle_asan.so" LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    ASAN_OPTIONS=detect_leaks=0 python -m pytest tests -q -m "not gpu" -p no:cacheprovider --deselect tests/test_abi_exceptions_cpu.py
rm -rf "$OBJ"
