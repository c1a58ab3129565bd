#!/bin/bash
# Host-side AddressSanitizer + UndefinedBehaviorSanitizer pass over libadfp's C ABI (tests/asan/host_driver.cpp explains what runs).
# usage: build_and_run.sh <output directory>
set -e
OUT=${1:-/tmp/adfp_asan}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
CLANGXX=${CLANGXX:-/opt/rocm/lib/llvm/bin/clang++}
mkdir -p "$OUT"
# the product's flags (csrc/build.sh) at -O1 -g, host translation instrumented, device code as in the product
$HIPCC -O1 -g --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fsanitize=address,undefined -fno-gpu-sanitize \
    -fno-sanitize-recover=undefined -I"$ROOT/include" -shared -fPIC -o "$OUT/libadfp_asan.so" "$ROOT/attentive_dfprior_amd/csrc/adfp_kernels.hip"
$CLANGXX -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -I"$ROOT/include" "$ROOT/tests/asan/host_driver.cpp" \
    -L"$OUT" -ladfp_asan -Wl,-rpath,"$OUT" -o "$OUT/host_driver"
# leak detection off: the HIP runtime keeps its own allocations for the life of the process
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 "$OUT/host_driver"
