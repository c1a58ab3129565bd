#!/bin/bash
# Timing-only sensitivity builds of the f16-split decoder (WRONG RESULTS by construction; never part of libadfp.so):
# each switches one phase of k_decode_h off so that its share of the launch can be read from tools/ab_stage.py.
#   build/libadfp_exp_<name>.so  for name in mfma2 noldsw nosin nogather nosplit
# Usage: bash tools/experiments/build_sensitivity.sh ; then on the GPU box
#   for n in mfma2 noldsw nosin nogather nosplit; do ADFP_LIB_PATH=$PWD/build/libadfp_exp_$n.so python tools/ab_stage.py; done
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
TMP=$(mktemp -d)
cp -r "$ROOT/attentive_dfprior_amd/csrc" "$TMP/csrc"
cp -r "$ROOT/include" "$TMP/include"
cd "$TMP/csrc"
python3 - <<'PY'
import re
p='adfp_decode_h.h'
s=open(p).read()
# third product off
s=s.replace('''        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh[ks], acc, 0, 0, 0);
    }''','''#ifndef EXP_MFMA2
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh[ks], acc, 0, 0, 0);
#endif
    }''')
# weights read once per chain instead of per k-step
s=s.replace('''        const f16x8 ah = __builtin_bit_cast(f16x8, *(const u32x4*)(w + ks * 512 + lane_off));
        const f16x8 al = __builtin_bit_cast(f16x8, *(const u32x4*)(w + ks * 512 + 256 + lane_off));''','''#ifdef EXP_NOLDSW
        const f16x8 ah = __builtin_bit_cast(f16x8, *(const u32x4*)(w + lane_off));
        const f16x8 al = __builtin_bit_cast(f16x8, *(const u32x4*)(w + 256 + lane_off));
#else
        const f16x8 ah = __builtin_bit_cast(f16x8, *(const u32x4*)(w + ks * 512 + lane_off));
        const f16x8 al = __builtin_bit_cast(f16x8, *(const u32x4*)(w + ks * 512 + 256 + lane_off));
#endif''')
s=s.replace('''                e[j] = adfp_sinf(arg);''','''#ifdef EXP_NOSIN
                e[j] = arg;
#else
                e[j] = adfp_sinf(arg);
#endif''')
s=s.replace('''            gather16(a.g0, pn, h, c);
            if (CDIM == 64) gather16(a.g1, pn, h, c + 16);''','''#ifdef EXP_NOGATHER
            for (int k = 0; k < CDIM / 2; ++k) c[k] = pn[0] * (float)(k + 1);
#else
            gather16(a.g0, pn, h, c);
            if (CDIM == 64) gather16(a.g1, pn, h, c + 16);
#endif''')
s=s.replace('''        const h2 hp = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a, b));
        h2 lp;
        lp[0] = (_Float16)__builtin_fmaf((float)hp[0], (float)neg1[0], a);
        lp[1] = (_Float16)__builtin_fmaf((float)hp[1], (float)neg1[0], b);''','''        const h2 hp = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a, b));
        h2 lp;
#ifdef EXP_NOSPLIT
        lp = hp;
#else
        lp[0] = (_Float16)__builtin_fmaf((float)hp[0], (float)neg1[0], a);
        lp[1] = (_Float16)__builtin_fmaf((float)hp[1], (float)neg1[0], b);
#endif''')
open(p,'w').write(s)
p='adfp_device.h'
s=open(p).read()
s=s.replace("ADFP_DEV float relu_f(float x) { const int b = __float_as_int(x); return __int_as_float(b > 0 ? b : 0); }","""#ifdef EXP_NORELU
ADFP_DEV float relu_f(float x) { return x; }
#elif defined(EXP_RELUADD)
ADFP_DEV float relu_f(float x) { return x + __builtin_fabsf(x); }
#else
ADFP_DEV float relu_f(float x) { const int b = __float_as_int(x); return __int_as_float(b > 0 ? b : 0); }
#endif""")
s=s.replace("""    const float k = fmaf(x, 0.15915494f, 12582912.0f) - 12582912.0f;
    float t = fmaf(x, 0.15915494f, -k);
    return fmaf(x, 6.4206382432985265e-09f, t);""","""    const float k = fmaf(x, 0.15915494f, 12582912.0f) - 12582912.0f;
    float t = fmaf(x, 0.15915494f, -k);
#ifdef EXP_TURN1
    return t;
#else
    return fmaf(x, 6.4206382432985265e-09f, t);
#endif""")
s=s.replace("""ADFP_DEV float adfp_turns(float x) {""","""#ifdef EXP_VCONST
ADFP_DEV float adfp_turns(float x) {
    float magic = 12582912.0f, clo = 6.4206382432985265e-09f;
    asm volatile("" : "+v"(magic), "+v"(clo));
    const float k = fmaf(x, 0.15915494f, magic) - magic;
    float t = fmaf(x, 0.15915494f, -k);
    return fmaf(x, clo, t);
}
#define adfp_turns adfp_turns_unused
#endif
ADFP_DEV float adfp_turns(float x) {""")
s=s.replace("ADFP_DEV float adfp_sinf(float x) { return __builtin_amdgcn_sinf(adfp_turns(x)); }","""#ifdef EXP_VCONST
#undef adfp_turns
#endif
ADFP_DEV float adfp_sinf(float x) { return __builtin_amdgcn_sinf(adfp_turns(x)); }""")
open(p,'w').write(s)
p='adfp_decode_h.h'
s=open(p).read()
s=s.replace("""#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = x[2 * j], b = x[2 * j + 1];
        if (CHECK) amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), amax);
        const h2 hp = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a, b));""","""#ifdef EXP_GUARDTREE
    if (CHECK) {
        const float m0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(x[0]), __builtin_fabsf(x[1])), __builtin_fabsf(x[2]));
        const float m1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(x[3]), __builtin_fabsf(x[4])), __builtin_fabsf(x[5]));
        const float m2 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(x[6]), __builtin_fabsf(x[7])), amax);
        amax = __builtin_fmaxf(__builtin_fmaxf(m0, m1), m2);
    }
#define EXP_NOGUARD 1
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = x[2 * j], b = x[2 * j + 1];
        if (CHECK) amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), amax);
        const h2 hp = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a, b));""")
s=s.replace("        if (CHECK) amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), amax);","#ifndef EXP_NOGUARD\n        if (CHECK) amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), amax);\n#endif")
open(p,'w').write(s)
PY
mkdir -p "$ROOT/build"
for n in ${VARIANTS:-mfma2 noldsw nosin nogather nosplit norelu reluadd turn1 noguard}; do
  D=$(echo $n | tr a-z A-Z)
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I../include -DEXP_$D -shared -fPIC \
      -o "$ROOT/build/libadfp_exp_$n.so" adfp_kernels.hip &
done
wait
rm -rf "$TMP"
ls -la "$ROOT/build"
