#!/bin/bash
# Debug build for tools/experiments/att_span.py: a COPY of csrc/ patched so that the attention network's training forward
# (k_attention_h<1, 256>) records, per workgroup, the wall clock (100 MHz) at its start, once its weight image is in LDS, after each of
# wave 0's tiles, at the end of wave 0's tile loop.  The product sources are not touched.  -> tools/ab_libs/libadfp_att_span.so
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
W=$(mktemp -d /tmp/att_span.XXXX)
cp -r "$ROOT/attentive_dfprior_amd/csrc" "$W/csrc"
python3 - "$W/csrc" <<'PY'
import sys
d = sys.argv[1]
p = d + '/adfp_decode_h.h'
s = open(p).read()
def rep(old, new):
    global s
    assert s.count(old) == 1, (old, s.count(old))
    s = s.replace(old, new)
rep('template <int TRAIN, int NT = 512>\n__global__ __launch_bounds__(NT) void k_attention_h(AttArgs a) {\n    using A = AttLayoutH;\n    using ST = AttStage;\n',
    '__device__ unsigned long long g_att_span[16 * 1024];\n'
    'template <int TRAIN, int NT = 512>\n__global__ __launch_bounds__(NT) void k_attention_h(AttArgs a) {\n    using A = AttLayoutH;\n    using ST = AttStage;\n'
    '    unsigned long long* span_ = g_att_span + 16 * (blockIdx.x < 1024 ? blockIdx.x : 1023); int ntile_ = 0;\n'
    '    const bool stamp_ = TRAIN && threadIdx.x == 0 && blockIdx.x < 1024;\n'
    '    if (stamp_) { for (int q = 0; q < 16; ++q) span_[q] = 0; span_[0] = wall_clock64(); }\n')
rep('    if (threadIdx.x == 0) s_next = NT / 64;\n    __syncthreads();\n    const float* lds = (const float*)ldsu;\n    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;\n    const int lane_off = h * 128 + p * 4;\n    const int count = a.count_ptr ? *a.count_ptr : a.n_rows;',
    '    if (threadIdx.x == 0) s_next = NT / 64;\n    __syncthreads();\n    if (stamp_) span_[1] = wall_clock64();\n    const float* lds = (const float*)ldsu;\n    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;\n    const int lane_off = h * 128 + p * 4;\n    const int count = a.count_ptr ? *a.count_ptr : a.n_rows;')
rep('    for (int j = threadIdx.x >> 6, tile; (tile = claim_tile<NT / 64>(j, &s_next, ntiles)) >= 0;) {\n        const int idx = tile * 32 + p;\n        const bool valid = idx < count;\n        const int ii = valid ? idx : 0;\n        const float occ = a.att_occ[ii], u = a.att_u[ii];',
    '    for (int j = threadIdx.x >> 6, tile; (tile = claim_tile<NT / 64>(j, &s_next, ntiles)) >= 0;) {\n        if (stamp_ && ntile_ < 6) span_[4 + 2 * ntile_] = wall_clock64();\n        ++ntile_;\n        const int idx = tile * 32 + p;\n        const bool valid = idx < count;\n        const int ii = valid ? idx : 0;\n        const float occ = a.att_occ[ii], u = a.att_u[ii];')
rep("            a.w[q] = a1;\n        }\n    }\n    report_range(a.status, amax, ADFP_STATUS_F16_RANGE_ATT, a.call_flag);\n}",
    "            a.w[q] = a1;\n        }\n        if (stamp_ && ntile_ <= 6) span_[3 + 2 * ntile_] = wall_clock64();\n    }\n"
    "    if (stamp_) { span_[2] = wall_clock64(); span_[3] = ntile_; }\n    report_range(a.status, amax, ADFP_STATUS_F16_RANGE_ATT, a.call_flag);\n}")
open(p, 'w').write(s)
p = d + '/adfp_kernels.hip'
s = open(p).read()
rep('#include "adfp_backward_fused.h"\n',
    '#include "adfp_backward_fused.h"\nextern "C" int adfp_debug_att_span(unsigned long long* host_out) {\n'
    '    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_att_span), sizeof(unsigned long long) * 16 * 1024);\n}\n')
open(p, 'w').write(s)
PY
mkdir -p "$ROOT/tools/ab_libs"
cd "$W/csrc"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I"$ROOT/include" -shared -fPIC \
    -o "$ROOT/tools/ab_libs/libadfp_att_span.so" adfp_kernels.hip
rm -rf "$W"
echo built att_span
