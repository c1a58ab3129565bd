#!/bin/bash
# Build libadfp.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
# -ffp-contract=off: HIP's __fmul_rn/__fadd_rn are plain * and + and would be fused into fma under the default
# contraction; the kernels reproduce the reference's separate roundings and use fmaf() explicitly where they want one.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -fno-slp-vectorize: the SLP vectoriser packs neighbouring scalar f32 ops into v_pk_fma_f32 / v_pk_add_f32, which cost MORE issue
# time than the two scalar instructions next to MFMAs on gfx950 (tools/micro/mfma_fill.hip: 4 cycles each + ~14 per MFMA gap);
# colour decoder 0.896 -> 0.868 ms per 100 000-ray batch.
$HIPCC -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I../../include -shared -fPIC \
    -o ../libadfp.so adfp_kernels.hip "$@"
