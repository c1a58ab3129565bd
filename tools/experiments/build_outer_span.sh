#!/bin/bash
# Debug build for tools/experiments/outer_span.py: a COPY of csrc/ patched so that k_outer_h records, per workgroup, the wall clock
# (100 MHz) at its start, after the first tile is in LDS, after each of its first 20 tiles, at the end of the tile loop and at its
# end.  The product sources are not touched (the profile stamps hash them).  -> tools/ab_libs/libadfp_outer_span.so
# With the argument "rows": the archived rewrite k_outer_rows.h on the attention path, stamped the same way -> libadfp_outer_span_rows.so
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
W=$(mktemp -d /tmp/outer_span.XXXX)
cp -r "$ROOT/attentive_dfprior_amd/csrc" "$W/csrc"
python3 - "$W/csrc" "$ROOT" "${1:-h}" <<'PY'
import sys
d = sys.argv[1]
root, variant = sys.argv[2], sys.argv[3]
p = d + '/adfp_backward_h.h'
s = open(p).read()
def rep(old, new, cnt=1):
    global s
    assert s.count(old) == cnt, (old, s.count(old))
    s = s.replace(old, new)
rep('__global__ __launch_bounds__(512) void k_outer_h(OuterHArgs b) {\n    const OuterArgs& a = b.o;',
    '__device__ unsigned long long g_outer_span[32 * 256];\n'
    '__global__ __launch_bounds__(512) void k_outer_h(OuterHArgs b) {\n    const OuterArgs& a = b.o;\n'
    '    unsigned long long* span_ = g_outer_span + 32 * blockIdx.x; int ntile_ = 0;\n'
    '    if (threadIdx.x == 0) { for (int q = 0; q < 32; ++q) span_[q] = 0; span_[0] = wall_clock64(); }')
rep('    float amax = 0.f;\n    fetch(m);\n    for (;;) {\n#pragma unroll\n        for (int rr = 0; rr < 2; ++rr) {                   // registers -> f32 tile',
    '    float amax = 0.f;\n    if (threadIdx.x == 0) span_[1] = wall_clock64();\n    fetch(m);\n    for (;;) {\n#pragma unroll\n        for (int rr = 0; rr < 2; ++rr) {                   // registers -> f32 tile')
rep('        if (!more) break;\n        m = nm; blk = nblk; m1 = nm1;\n    }\n    if (!(b.skip && *b.skip)) report_range(b.status, amax, ADFP_STATUS_F16_RANGE_BWD);\n    float* part = a.partial + (long long)blockIdx.x * a.part_stride;',
    '        if (threadIdx.x == 0 && ntile_ < 20) span_[4 + ntile_] = wall_clock64();\n        ++ntile_;\n'
    '        if (!more) break;\n        m = nm; blk = nblk; m1 = nm1;\n    }\n'
    '    __syncthreads();\n    if (threadIdx.x == 0) { span_[2] = wall_clock64(); span_[24] = ntile_; }\n'
    '    if (!(b.skip && *b.skip)) report_range(b.status, amax, ADFP_STATUS_F16_RANGE_BWD);\n    float* part = a.partial + (long long)blockIdx.x * a.part_stride;')
if variant == 'rows':            # the archived rewrite (tools/experiments/k_outer_rows.h) in place of k_outer_h on the attention path, stamped the same way
    k = open(root + '/tools/experiments/k_outer_rows.h').read()
    k = k[k.index('template <int NC, int JW>'):]
    def krep(old, new):
        global k
        assert k.count(old) == 1, old
        k = k.replace(old, new)
    krep('    const OuterArgs& a = b.o;\n', '    const OuterArgs& a = b.o;\n    unsigned long long* span_ = g_outer_span + 32 * blockIdx.x; int ntile_ = 0;\n'
         '    if (threadIdx.x == 0) { for (int q = 0; q < 32; ++q) span_[q] = 0; span_[0] = wall_clock64(); }\n')
    krep('    fetch(ld0, true);\n', '    if (threadIdx.x == 0) span_[1] = wall_clock64();\n    fetch(ld0, true);\n')
    k = k.replace('        products(st[0]);\n', '        products(st[0]);\n        if (threadIdx.x == 0 && ntile_ < 20) span_[4 + ntile_] = wall_clock64();\n        ++ntile_;\n')
    k = k.replace('        products(st[1]);\n', '        products(st[1]);\n        if (threadIdx.x == 0 && ntile_ < 20) span_[4 + ntile_] = wall_clock64();\n        ++ntile_;\n')
    krep('    if (!(b.skip && *b.skip)) report_range(b.status, amax, ADFP_STATUS_F16_RANGE_BWD);\n',
         '    __syncthreads();\n    if (threadIdx.x == 0) { span_[2] = wall_clock64(); span_[24] = ntile_; }\n'
         '    if (!(b.skip && *b.skip)) report_range(b.status, amax, ADFP_STATUS_F16_RANGE_BWD);\n')
    j2 = k.rindex('}\n')
    k = k[:j2] + '    __builtin_amdgcn_s_waitcnt(0); __syncthreads();\n    if (threadIdx.x == 0) span_[3] = wall_clock64();\n}\n\n'
    tail0 = '// flat[e] += 2^-k sum over the workgroup slots of partial[slot][e]'
    s = s.replace(tail0, k + tail0)
# the kernel's last statement: the epilogue loop closes with "        }\n    }\n}\n" right before the reduce kernel's comment
tail = '// flat[e] += 2^-k sum over the workgroup slots of partial[slot][e]'
i = s.index(tail)
j = s.rindex('}\n', 0, i)
s = s[:j] + '    __builtin_amdgcn_s_waitcnt(0); __syncthreads();\n    if (threadIdx.x == 0) span_[3] = wall_clock64();\n}\n' + s[j + 2:]
open(p, 'w').write(s)
p = d + '/adfp_kernels.hip'
s = open(p).read()
rep('#include "adfp_backward_fused.h"\n',
    '#include "adfp_backward_fused.h"\nextern "C" int adfp_debug_outer_span(unsigned long long* host_out) {\n'
    '    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_outer_span), sizeof(unsigned long long) * 32 * 256);\n}\n')
if variant == 'rows':
    rep('                    hipLaunchKernelGGL(k_outer_h, dim3(nblk < OUTER_NSLOT ? nblk : OUTER_NSLOT), dim3(512), 0, st, oh);',
        '                    hipLaunchKernelGGL((k_outer_rows<AttStage::NCOLS, 7>), dim3(nblk < OUTER_NSLOT ? nblk : OUTER_NSLOT), dim3(512), 0, st, oh);')
open(p, 'w').write(s)
PY
mkdir -p "$ROOT/tools/ab_libs"
cd "$W/csrc"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I"$ROOT/include" -shared -fPIC \
    -o "$ROOT/tools/ab_libs/libadfp_outer_span${1:+_$1}.so" adfp_kernels.hip
rm -rf "$W"
echo built outer_span
