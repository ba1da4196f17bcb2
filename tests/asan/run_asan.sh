#!/bin/bash
# Host code of libpnmol_hip under AddressSanitizer + UndefinedBehaviorSanitizer, on a machine WITHOUT a GPU: the library is
# built with host-only instrumentation (device code untouched), the HIP runtime is replaced by tests/asan/hip_mock.c
# (LD_PRELOAD), tests/asan/driver.cpp walks the C ABI.  ~3 minutes (two sanitizer builds of the .hip files).
# Usage: tests/asan/run_asan.sh [logfile]     exit code 0 = sanitizers silent and every return code as expected.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="${TMPDIR:-/tmp}/pnmol_asan"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
CLANG="/opt/rocm/lib/llvm/bin/clang++"
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
"$HIPCC" --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared $SAN -fno-gpu-sanitize -shared-libsan -Wno-unused-value -Wno-unused-result \
    -I"$ROOT/include" "$ROOT/pnmol-experiments_amd/csrc/pnmol_hip.hip" "$ROOT/pnmol-experiments_amd/csrc/pnmol_sqrt.hip" -o "$OUT/libpnmol_asan.so"
gcc -O1 -g -fPIC -shared "$ROOT/tests/asan/hip_mock.c" -o "$OUT/libhipmock.so"
"$CLANG" -O1 -g -std=c++17 $SAN -shared-libsan -I"$ROOT/include" "$ROOT/tests/asan/driver.cpp" -L"$OUT" -lpnmol_asan -Wl,-rpath,"$OUT" -o "$OUT/asan_driver"
RT="$(dirname "$("$CLANG" -print-file-name=libclang_rt.asan-x86_64.so)")"
export LD_LIBRARY_PATH="$RT:/opt/rocm/lib:${LD_LIBRARY_PATH:-}"
export ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1" UBSAN_OPTIONS="print_stacktrace=1"
LD_PRELOAD="$RT/libclang_rt.asan-x86_64.so:$OUT/libhipmock.so" "$OUT/asan_driver" 2>&1 | tee "${1:-$OUT/asan.log}"
exit "${PIPESTATUS[0]}"
