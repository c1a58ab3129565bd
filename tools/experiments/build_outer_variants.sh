#!/bin/bash
# Timing-only variants of k_outer_h (WRONG RESULTS by construction; never part of libadfp.so):
#   build/libadfp_exp_outer_hot.so     every tile re-reads the workgroup's FIRST tile (cache-hot): what the tile costs without HBM
#   build/libadfp_exp_outer_nomath.so  loads + LDS stores + barriers only: what the load pipeline alone sustains
# Usage: bash tools/experiments/build_outer_variants.sh; on the GPU box
#   export ADFP_LIB_PATH=$PWD/build/libadfp_exp_outer_hot.so; rocprofv3 --kernel-trace --stats ... -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
TMP=$(mktemp -d)
cp -r "$ROOT/attentive_dfprior_amd/csrc" "$TMP/csrc"
cp -r "$ROOT/include" "$TMP/include"
cd "$TMP/csrc"
python3 - <<'PY'
p='adfp_backward_h.h'
s=open(p).read()
old="        if (more) { m1 = nm1; fetch(nm); m1 = cur_m1; }    // in flight during the conversion and the MFMAs"
assert old in s
s=s.replace(old,'''#ifdef EXP_OUTER_HOT
        if (more) { const int keep_m1 = m1; m1 = first_m + OUTER_RT; fetch(first_m); m1 = keep_m1; }
#else
        if (more) { m1 = nm1; fetch(nm); m1 = cur_m1; }    // in flight during the conversion and the MFMAs
#endif
#ifdef EXP_OUTER_NOMATH
        if (!more) break;
        m = nm; blk = nblk; m1 = nm1;
        continue;
#endif''')
old="    float amax = 0.f;\n    fetch(m);\n    for (;;) {"
assert old in s
s=s.replace(old,"    float amax = 0.f;\n    const int first_m = m;\n    fetch(m);\n    for (;;) {")
open(p,'w').write(s)
PY
mkdir -p "$ROOT/build"
for v in HOT NOMATH; do
    n=$(echo $v | tr A-Z a-z)
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DEXP_OUTER_$v -I../include -shared -fPIC \
        -o "$ROOT/build/libadfp_exp_outer_$n.so" adfp_kernels.hip &
done
wait
rm -rf "$TMP"
ls -la "$ROOT/build/" | grep outer
