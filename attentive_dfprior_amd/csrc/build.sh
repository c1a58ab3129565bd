#!/bin/bash
# Build libadfp.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
# -ffp-contract=off: HIP's __fmul_rn/__fadd_rn are plain * and + and would be fused into fma under the default
# contraction; the kernels reproduce the reference's separate roundings and use fmaf() explicitly where they want one.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -I../../include -shared -fPIC \
    -o ../libadfp.so adfp_kernels.hip "$@"
