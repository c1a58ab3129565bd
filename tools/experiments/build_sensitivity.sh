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
PY
mkdir -p "$ROOT/build"
for n in mfma2 noldsw nosin nogather nosplit; do
  D=$(echo $n | tr a-z A-Z)
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I../include -DEXP_$D -shared -fPIC \
      -o "$ROOT/build/libadfp_exp_$n.so" adfp_kernels.hip &
done
wait
rm -rf "$TMP"
ls -la "$ROOT/build"
