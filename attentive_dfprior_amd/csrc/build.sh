#!/bin/bash
# Build libadfp.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -O3 --offload-arch=gfx950 -std=c++17 -I../../include -shared -fPIC \
    -o ../libadfp.so adfp_kernels.hip "$@"
