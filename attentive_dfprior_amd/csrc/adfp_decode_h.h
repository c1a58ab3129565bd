// adfp_decode_h.h -- decoder kernel with the MLP on v_mfma_f32_32x32x16_f16 using a 3-product
// split of every f32 operand:  a = a_hi + a_lo (a_hi = a truncated to 11 significant bits, a_lo the
// f16-rounded remainder),  a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  accumulated in fp32.
//
// Why.  Measured on MI355X (tools/micro/mfma_valu_*.hip): VALU instructions barely hide behind MFMAs on
// a SIMD -- next to a dependent f32 MFMA chain a VALU-only partner wave gets ~2 % of its issue rate, next to
// an f16 MFMA stream ~2 instructions per 32-cycle slot, and a wave's own fillers cost 2.5-4 cycles each.
// The exact-f32 decoder therefore costs  64 cyc x 240 f32 MFMAs + the VALU work  per 32-point tile, the MFMA
// term is 70 % of it and f32-input MFMA runs at 1/16 of the f16 rate.  Three f16 MFMAs per 16 k-values
// replace eight f32 MFMAs: 90 x 32 cycles instead of 240 x 64.  Per-phase stamps of this kernel (debug build
// -DADFP_STAMPS, tools/ab_stage.py): gather 32 %, Fourier features 25 %, the five layers 33 %, output layer +
// store 8 % of the wave-cycles.
//
// Accuracy (tools/micro/f16x3_accuracy.hip, K = 128): max error relative to the largest output
// 3.6e-7 for O(1) operands and 9.5e-7 for O(0.03) operands, against 3.6e-7 / 2.5e-7 for the exact f32
// MFMA; f16 products are exact in fp32 and f16 subnormal inputs are not flushed on gfx950.  The
// dropped a_lo*b_lo term is 2^-22 relative.  Operands must stay below 65504 in magnitude (hidden
// activations of this network are O(1..10)); every value that is split is range-checked and a violation raises
// the sticky ADFP_STATUS_F16_RANGE bit of adfp_scene.status instead of passing silently.  The exact-f32
// kernel (k_decode) stays available (ADFP_MATH=f32) and is what the backward uses.
#pragma once
#include "adfp_device.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// "H" image of one decoder: Fourier matrix / biases / output layer in f32, weight chains in f16.
// A chain block [32 out x K in] is K/16 k-steps; a k-step is [hi|lo][h][32 rows][8 j] halves
// (1024 halves = 2 KB): lane (i,h) reads its 8 hi and 8 lo halves with two ds_read_b128.
// k-step ks, lane-half h, element j carries unit 32*(ks>>1) + kmapH(8*(ks&1) + j, h): the 16
// accumulator registers of a layer, split in two groups of 8, are the next layer's two k-steps.
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr int unit_of_h(int ks, int h, int j) { return 32 * (ks >> 1) + kmapH(8 * (ks & 1) + j, h); }

template <int CDIM, int NOUT>
struct DecLayoutH {
    using F = DecLayout<CDIM, NOUT>;                 // flat (state_dict) offsets are shared
    static constexpr int KS_E = 6;                   // k-steps of the 96 (93) Fourier features
    static constexpr int KS_C = CDIM / 16;
    __host__ __device__ static constexpr int ks(int i) { return i == 0 ? KS_E : (i == 3 ? KS_E + 2 : 2); }
    // offsets in 32-bit words (a k-step = 512 words)
    static constexpr int P_BM = 0;                                   // [96][4] f32
    __host__ __device__ static constexpr int layer_words(int i) { return ks(i) * 512 + 32 + KS_C * 512 + 32; }
    __host__ __device__ static constexpr int P_WP(int i) {
        int o = 384;
        for (int k = 0; k < i; ++k) o += layer_words(k);
        return o;
    }
    __host__ __device__ static constexpr int P_BP(int i) { return P_WP(i) + ks(i) * 512; }
    __host__ __device__ static constexpr int P_WC(int i) { return P_BP(i) + 32; }
    __host__ __device__ static constexpr int P_BC(int i) { return P_WC(i) + KS_C * 512; }
    static constexpr int P_WO = P_WP(5);                              // [2][NOUT][16] f32
    static constexpr int P_BO = P_WO + 2 * NOUT * 16;
    // one word per 256-word block of the pack kernel: non-zero = that block met a weight outside the f16 range (image_out_of_range)
    static constexpr int P_FLAG = P_BO + 4;
    static constexpr int NFLAG = (((P_FLAG + 511) / 256) + 3) & ~3;
    static constexpr int P_TOTAL = P_FLAG + NFLAG;
    static_assert((P_TOTAL + 255) / 256 <= NFLAG, "one flag word per pack block");
};

// source of 32-bit word t of the H image: either one f32 of the flat buffer (kind 0) or a pair of
// halves (kind 1 = hi parts, kind 2 = lo parts) of two flat weights
struct HSrc { int kind, s0, s1; };
template <int CDIM, int NOUT>
__device__ HSrc dec_h_src(int t) {
    using L = DecLayoutH<CDIM, NOUT>;
    using F = DecLayout<CDIM, NOUT>;
    if (t < 384) {
        const int j = t >> 2, c = t & 3;
        return HSrc{0, (j < 93 && c < 3) ? F::F_EB + c * 93 + j : -1, -1};
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        if (t < L::P_BP(i) || (t >= L::P_WC(i) && t < L::P_BC(i))) {
            const bool fc = t >= L::P_WC(i);
            const int u = t - (fc ? L::P_WC(i) : L::P_WP(i));
            const int ks = u >> 9, part = (u >> 8) & 1, h = (u >> 7) & 1, row = (u >> 2) & 31, jp = (u & 3) * 2;
            int src[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int col = unit_of_h(ks, h, jp + e);
                if (fc) src[e] = F::F_FC(i) + row * CDIM + col;
                else {
                    if (i == 0) { if (col >= 93) col = -1; }
                    else if (i == 3) { if (col < 96) { if (col >= 93) col = -1; } else col = 93 + (col - 96); }
                    src[e] = col < 0 ? -1 : F::F_PL(i) + row * F::in_dim(i) + col;
                }
            }
            return HSrc{1 + part, src[0], src[1]};
        }
        if (t < L::P_WC(i)) return HSrc{0, F::F_PL(i) + 32 * F::in_dim(i) + (t - L::P_BP(i)), -1};
        if (t < L::P_BC(i) + 32) return HSrc{0, F::F_FC(i) + 32 * CDIM + (t - L::P_BC(i)), -1};
    }
    if (t < L::P_BO) {
        const int u = t - L::P_WO;
        const int h = u / (NOUT * 16), o = (u >> 4) % NOUT, r = u & 15;
        return HSrc{0, F::F_OW + o * 32 + kmapH(r, h), -1};
    }
    const int o = t - L::P_BO;
    return HSrc{0, o < NOUT ? F::F_OB + o : -1, -1};
}

__device__ __forceinline__ float f16_hi_part(float x) { return __uint_as_float(__float_as_uint(x) & 0xFFFFE000u); }
__device__ __forceinline__ float f16_clamp(float x) { return fminf(fmaxf(x, -65504.0f), 65504.0f); }

// Range flags of an H image: every workgroup of the pack kernel (NFLAG of them, one per 256 words; the grid is exactly NFLAG blocks)
// ORs what its own threads saw and stores ONE word -- plain stores, nothing to zero first, no serial sweep over the parameters.
// The decoder kernels OR the NFLAG words once per workgroup while they wait for their LDS image anyway.
template <int P_FLAG, int NFLAG>
ADFP_DEV bool image_out_of_range(const unsigned* ldsu) {
    const int bad = (int)threadIdx.x < NFLAG ? (int)ldsu[P_FLAG + threadIdx.x] : 0;
    return __syncthreads_or(bad) != 0;
}
ADFP_DEV void pack_block_flag(bool bad, unsigned* __restrict__ flag_word, int* __restrict__ status, int bit) {
    const int any = __syncthreads_or(bad ? 1 : 0);
    if (threadIdx.x == 0) {
        *flag_word = any ? 1u : 0u;
        if (any && status) __hip_atomic_fetch_or(status, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
template <int CDIM, int NOUT>
ADFP_DEV void pack_decoder_h_block(int blk, const float* __restrict__ flat, unsigned* __restrict__ packed, int* __restrict__ status, int bit) {
    using L = DecLayoutH<CDIM, NOUT>;
    const int t = blk * 256 + (int)threadIdx.x;
    HSrc s{0, -1, -1};
    float a = 0.f, b = 0.f;
    if (t < L::P_FLAG) {
        s = dec_h_src<CDIM, NOUT>(t);
        a = s.s0 < 0 ? 0.f : flat[s.s0]; b = s.s1 < 0 ? 0.f : flat[s.s1];
    }
    pack_block_flag(!(fmaxf(fabsf(a), fabsf(b)) < 65504.0f), packed + L::P_FLAG + blk, status, bit);      // every f32 word and every weight passes through here once
    if (t >= L::P_FLAG) return;
    if (s.kind == 0) { packed[t] = s.s0 < 0 ? 0u : __float_as_uint(a); return; }
    a = f16_clamp(a); b = f16_clamp(b);          // out of range (flagged above / by pack_range_flag): stay finite, 0 x inf must not appear downstream
    const float ah = f16_hi_part(a), bh = f16_hi_part(b);
    _Float16 x, y;
    if (s.kind == 1) { x = (_Float16)ah; y = (_Float16)bh; }
    else { x = (_Float16)(a - ah); y = (_Float16)(b - bh); }
    packed[t] = (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
template <int CDIM, int NOUT>
__global__ void k_pack_decoder_h(const float* __restrict__ flat, unsigned* __restrict__ packed, int* __restrict__ status, int bit) { pack_decoder_h_block<CDIM, NOUT>((int)blockIdx.x, flat, packed, status, bit); }

// 8 f32 -> 8 hi halves + 8 lo halves.  `amax` tracks max |x| of everything that was split (one v_max3_f32 per
// pair): an operand at or beyond the f16 range (65504) cannot be split -- cvt_pkrtz saturates it -- so the kernels
// raise the sticky ADFP_STATUS_F16_RANGE bit of adfp_scene.status when amax reaches it (CHECK = false for values
// that are bounded by construction: sines).
#ifdef ADFP_SPLIT_MASK
// 3 VALU per value: and, sub, 2 x cvt_pkrtz per pair.  hi = x truncated to 11 significant bits (a mask;
// exactly representable in f16), lo = x - hi.
template <bool CHECK = true>
ADFP_DEV void split8(const float* __restrict__ x, f16x8& hi, f16x8& lo, float& amax) {
    u32x4 uh, ul;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = x[2 * j], b = x[2 * j + 1];
        if (CHECK) amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), amax);
        const float ah = f16_hi_part(a), bh = f16_hi_part(b);
        uh[j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ah, bh));
        ul[j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a - ah, b - bh));
    }
    hi = __builtin_bit_cast(f16x8, uh);
    lo = __builtin_bit_cast(f16x8, ul);
}
#else
// 1.5 VALU per value: hi = cvt_pkrtz (round toward zero = the 11-bit truncation) for a pair, then lo = x - hi as ONE
// v_fma_mixlo_f16 / v_fma_mixhi_f16 per value: the f16 half is widened inside the instruction, the difference is formed exactly
// in f32 and lands, rounded to f16, in the low / high half of the packed register -- no separate conversion of the remainders
// (round 2 spent a v_fma_mix_f32 per value plus a v_cvt_pkrtz per pair on them: 16 instead of 12 quarter-rate instructions per 8
// values; the split is 128 values per tile).
template <bool CHECK = true>
ADFP_DEV void split8(const float* __restrict__ x, f16x8& hi, f16x8& lo, float& amax) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    u32x4 uh, ul;
    unsigned m1 = 0xBC00BC00u;                       // f16 (-1, -1), opaque to the optimiser so that the
    asm volatile("" : "+v"(m1));                     // fma below stays fma(fpext, fpext, f32) = v_fma_mix*
    const h2 neg1 = __builtin_bit_cast(h2, m1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = x[2 * j], b = x[2 * j + 1];
        if (CHECK) amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), amax);
        const h2 hp = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a, b));
        h2 lp;
        lp[0] = (_Float16)__builtin_fmaf((float)hp[0], (float)neg1[0], a);
        lp[1] = (_Float16)__builtin_fmaf((float)hp[1], (float)neg1[0], b);
        uh[j] = __builtin_bit_cast(unsigned, hp);
        ul[j] = __builtin_bit_cast(unsigned, lp);
    }
    hi = __builtin_bit_cast(f16x8, uh);
    lo = __builtin_bit_cast(f16x8, ul);
    // pin the running maximum here: left free, the optimiser defers the v_max3 chain to the end of the tile and keeps (spills)
    // the f32 values it still has to look at
    if (CHECK) asm volatile("" : "+v"(amax));
}
#endif
#define ADFP_F16_MAX 65504.0f
// once per wave, after its tile loop: the sticky word for the host (which network, so that it can switch that one to the exact
// image) and the CALL's flag (device memory, zeroed by the call) that arms the f32 fallback kernel and the backward's gate
ADFP_DEV void report_range(int* status, float amax, int bit, int* call_flag = nullptr) {
    if (!(amax < ADFP_F16_MAX)) {
        if (status) __hip_atomic_fetch_or(status, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (call_flag) __hip_atomic_fetch_or(call_flag, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// NK k-steps of a chain: acc += W[:, units of k-step] * x   with the 3-product split
template <int NK>
ADFP_DEV void mfma_chain_h(f32x16& acc, const unsigned* __restrict__ w, int lane_off, const f16x8* __restrict__ xh, const f16x8* __restrict__ xl) {
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const f16x8 ah = __builtin_bit_cast(f16x8, *(const u32x4*)(w + ks * 512 + lane_off));
        const f16x8 al = __builtin_bit_cast(f16x8, *(const u32x4*)(w + ks * 512 + 256 + lane_off));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh[ks], acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting the next chain's LDS reads
}

#ifdef ADFP_STAMPS
__device__ unsigned long long g_stamps[2 * 8192];     // debug build only: per-wave start/end wall clock (100 MHz)
__device__ unsigned long long g_phase[8];             // debug build only: wave-cycles per tile phase, summed over waves
#define ADFP_PHASE(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = clock64(); \
                           __builtin_amdgcn_sched_barrier(0); ph_[k] += now_ - last_; last_ = now_; } while (0)
#else
#define ADFP_PHASE(k) do {} while (0)
#endif

// relu + the fc_c bias, recording which units are active: bit (15 - r) of the low half of `m` = (relu(acc[r]) > 0), the
// predicate of torch's relu backward.  Two instructions per unit (0 - bits has its sign bit set exactly for bits > 0;
// v_alignbit shifts it in).
ADFP_DEV void relu_bias_mask(f32x16& acc, const float* __restrict__ bias, int h, unsigned& m) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 t = *(const f32x4*)(bias + 8 * q + 4 * h);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v = relu_f(acc[4 * q + k]);
            m = __builtin_amdgcn_alignbit(m, 0u - __float_as_uint(v), 31);
            acc[4 * q + k] = v + t[k];
        }
    }
}

// TRAIN = 1 (masks only) / 2 (masks + layer inputs): the training forward, 2 waves per SIMD (at the 168 registers of 3 waves the
// mask words alone spill 55).  Besides its outputs the kernel leaves what the f16-split backward (adfp_backward_h.h)
// needs, so that nothing is recomputed there: the ReLU masks of the five layers (a.masks: 3 words per lane half, always) and,
// when the network's weight gradients are wanted (a.act != NULL), the inputs of every layer -- position (the Fourier features
// are recomputed from it), grid features, h_0..h_4 -- as the X piece of the point's staging row (DecStage: NXM floats per point).  A NaN position
// (a ray the Mapper's pre-filter drops) is decoded at the origin instead, so that the staged activations stay finite; its
// outputs are NaN as before.
template <int CDIM, int NOUT, int ROLE, int NT, int TRAIN = 0>
__global__ __launch_bounds__(NT, (NT >= 512 ? NT / 256 : (ROLE == ROLE_HIGH ? 1 : 2))) void k_decode_h(DecodeArgs a) {
    using L = DecLayoutH<CDIM, NOUT>;
    using ST = DecStage<CDIM>;
#ifdef ADFP_STAMPS
    const unsigned long long stamp0 = wall_clock64();
#endif
    __shared__ __attribute__((aligned(16))) unsigned ldsu[L::P_TOTAL];
    __shared__ int s_next;
    image_to_lds<NT, L::P_TOTAL / 4>(ldsu, a.packed);
    if (threadIdx.x == 0) s_next = NT / 64;
    __syncthreads();
    const float* lds = (const float*)ldsu;

    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
    const int lane_off = h * 128 + p * 4;            // words: [h][32 rows][4 words = 8 halves]
    const int count = (ROLE == ROLE_HIGH && a.count_ptr) ? *a.count_ptr : a.P.n;
    const int ntiles = (count + 31) >> 5;

#ifdef ADFP_STAMPS
    unsigned long long ph_[6] = {0, 0, 0, 0, 0, 0}, last_ = clock64();
#endif
    float amax = image_out_of_range<L::P_FLAG, L::NFLAG>(ldsu) ? INFINITY : 0.f;      // max |operand| this wave has split (f16 range guard); a weight out of range
    for (int j = threadIdx.x >> 6, tile; (tile = claim_tile<NT / 64>(j, &s_next, ntiles)) >= 0;) {
        ADFP_PHASE(0);                                  // ticket + loop overhead
        const int idx = tile * 32 + p;
        const bool valid = idx < count;
        int q = valid ? idx : 0;
        if (ROLE == ROLE_HIGH && a.list) q = a.list[q];       // list == NULL: the decoder alone over every point (MLP.forward)

        double pt[3]; float pn[3], pf[3];
        load_point(a.P, q, pt);
        normalize3(a.nb, pt, pn);
        pf[0] = (float)pt[0]; pf[1] = (float)pt[1]; pf[2] = (float)pt[2];   // p.float() decoder.py:189
        float* srow = nullptr;                          // TRAIN: the X part of this point's staging row
        if constexpr (TRAIN) {
            const bool pnan = (pt[0] != pt[0]) | (pt[1] != pt[1]) | (pt[2] != pt[2]);
            if (pnan) { pf[0] = 0.f; pf[1] = 0.f; pf[2] = 0.f; }
            // (the row's head [x, y, z, 1, 0 ...] is not stored: the backward rebuilds it from the point, OuterHArgs.x_skip4)
            if (TRAIN == 2 && a.act && valid) srow = a.act + (long long)(ROLE == ROLE_HIGH ? idx : q) * ST::NXM;
        }

        // grid features -> split halves (k-steps of fc_c)
        f16x8 ch[L::KS_C], cl[L::KS_C];
        {
            float c[CDIM / 2];
            gather16(a.g0, pn, h, c);
            if (CDIM == 64) gather16(a.g1, pn, h, c + 16);
            if constexpr (TRAIN) if (srow) {
                stage_block(srow, ST::xm(ST::SC), h, c, 0);
                if (CDIM == 64) stage_block(srow, ST::xm(ST::SC + 32), h, c, 16);
            }
#pragma unroll
            for (int ks = 0; ks < L::KS_C; ++ks) split8(c + 8 * ks, ch[ks], cl[ks], amax);
        }
        ADFP_PHASE(1);                                  // point, normalise, gather, split c
        // Fourier features sin(p @ B) (decoder.py:26-30) -> split halves (6 k-steps)
        f16x8 eh[L::KS_E], el[L::KS_E];
#pragma unroll
        for (int ks = 0; ks < L::KS_E; ++ks) {
            float e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 bm = *(const f32x4*)(lds + L::P_BM + unit_of_h(ks, h, j) * 4);
                const float arg = fmaf(pf[2], bm.z, fmaf(pf[1], bm.y, pf[0] * bm.x));
                e[j] = adfp_sinf(arg);
            }
            split8<false>(e, eh[ks], el[ks], amax);     // |sin| <= 1
        }
        ADFP_PHASE(2);                                  // Fourier features

        __builtin_amdgcn_sched_barrier(0);
        // h = relu(W_i h + b_i) + (Wc_i c + bc_i); skip-concat [emb, h] feeds layer 3 (decoder.py:192-199)
        f32x16 acc;
        f16x8 hh[2], hl[2];
        unsigned mk[5] = {0u, 0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            bias_init(acc, lds + L::P_BP(i), h);
            if (i == 0) mfma_chain_h<L::KS_E>(acc, ldsu + L::P_WP(0), lane_off, eh, el);
            else if (i == 3) {
                mfma_chain_h<L::KS_E>(acc, ldsu + L::P_WP(3), lane_off, eh, el);
                mfma_chain_h<2>(acc, ldsu + L::P_WP(3) + L::KS_E * 512, lane_off, hh, hl);
            } else mfma_chain_h<2>(acc, ldsu + L::P_WP(i), lane_off, hh, hl);
            if constexpr (TRAIN) relu_bias_mask(acc, lds + L::P_BC(i), h, mk[i]);
            else relu_bias(acc, lds + L::P_BC(i), h);
            mfma_chain_h<L::KS_C>(acc, ldsu + L::P_WC(i), lane_off, ch, cl);
            if constexpr (TRAIN) if (srow) stage_block(srow, ST::xm(ST::SH(i)), h, acc);
            if (i < 4) {
                float t[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = acc[r];
                split8(t, hh[0], hl[0], amax);
                split8(t + 8, hh[1], hl[1], amax);
            }
        }

        ADFP_PHASE(3);                                  // 5 layers
        // output_linear on the VALU in f32: each half holds 16 of the 32 hidden units
        float out[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const float* wo = lds + L::P_WO + (h * NOUT + o) * 16;
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s = fmaf(acc[r], wo[r], s);
            s += __shfl_xor(s, 32);
            out[o] = s + lds[L::P_BO + o];
        }

        nan_point_outputs<NOUT>(pt, out);
        if constexpr (TRAIN) if (valid) {
            unsigned* mrow = a.masks + ((long long)(ROLE == ROLE_HIGH ? idx : q) * 2 + h) * 3;
            mrow[0] = (mk[0] & 0xFFFFu) | (mk[1] << 16); mrow[1] = (mk[2] & 0xFFFFu) | (mk[3] << 16); mrow[2] = mk[4] & 0xFFFFu;
        }
        if (valid && h == 0) {
            if constexpr (ROLE == ROLE_LOW) {
                const bool inb = in_bound(pt, a.b);
                const unsigned f = a.flags ? a.flags[q] : 0u;
                a.raw[4ll * q + 3] = ((f & ADFP_F_BAND) || inb || !a.apply_bound) ? out[0] : 100.f;   // Renderer.py:64
                if (a.write_w) a.w[q] = 1.f;
            } else if constexpr (ROLE == ROLE_COLOR) {
                a.raw[4ll * q + 0] = out[0]; a.raw[4ll * q + 1] = out[1]; a.raw[4ll * q + 2] = out[2];
                if (a.single) a.raw[4ll * q + 3] = out[3];                          // MLP.forward of the colour decoder: all 4 outputs
            } else {
                a.att_occ[idx] = a.single ? out[0] : out[0] + a.raw[4ll * q + 3];    // high + low, decoder.py:342
            }
        }
        ADFP_PHASE(4);                                  // output layer + store
    }
    report_range(a.status, amax, ROLE == ROLE_LOW ? ADFP_STATUS_F16_RANGE_LOW : (ROLE == ROLE_HIGH ? ADFP_STATUS_F16_RANGE_HIGH : ADFP_STATUS_F16_RANGE_COLOR),
                 a.call_flag);
#ifdef ADFP_STAMPS
    if (lane == 0) for (int k = 0; k < 5; ++k) atomicAdd(&g_phase[k], ph_[k]);
    const int wave = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (lane == 0 && wave < 8192) { g_stamps[2 * wave] = stamp0; g_stamps[2 * wave + 1] = wall_clock64(); }
#endif
}

// =============================================================================================
// LOW + COLOR in ONE launch (stage color, inference).  Both decoders run on every sample point of a ray batch with the same
// position; as two launches each of them reconstructed the point (f64 o + d z), normalised it (f64), took its ticket, loaded its
// z_vals and stored its part of the 16-byte raw row on its own (partial-line stores from two kernels: 20.6 + 16.7 B written per
// sample for 16 B of payload, profiles/r02_pmc_hbm_traffic.csv).  Here a wave does the point work once per tile, evaluates the
// low network and then the colour network out of two weight images that share the CU's LDS (65 + 65 KB of 160 KB; still one
// 768-thread workgroup per CU, 3 waves per SIMD) and writes raw as ONE 16-byte store per point -- and a frame has one launch
// tail per batch instead of two.  decode_net_h is k_decode_h's per-tile network evaluation (32-channel grids, no training
// state), kept textually parallel to it.
// =============================================================================================
template <int NOUT>
ADFP_DEV void decode_net_h(const unsigned* __restrict__ ldsu, const GridDev& g, const float pn[3], const float pf[3], int h, int lane_off,
                           float& amax, float* __restrict__ out) {
    // `ldsu` = the workgroup's LDS array + this network's image offset, an opaque per-tile register value at the call site (see
    // k_decode_lc).  Every LDS access below is one of four lane-dependent bases plus an immediate below 64 KB: the weight rows
    // (lane_off), the bias rows (4 h), the Fourier rows (16 h) and the output rows (16 NOUT h).
    using L = DecLayoutH<32, NOUT>;
    const unsigned* wl = ldsu + lane_off;
    const float* bh = (const float*)ldsu + 4 * h;
    const float* b16 = (const float*)ldsu + 16 * h;
    const float* bw = (const float*)ldsu + 16 * NOUT * h;
    f16x8 ch[L::KS_C], cl[L::KS_C];
    {
        float c[16];
        gather16(g, pn, h, c);
#pragma unroll
        for (int ks = 0; ks < L::KS_C; ++ks) split8(c + 8 * ks, ch[ks], cl[ks], amax);
    }
    f16x8 eh[L::KS_E], el[L::KS_E];
#pragma unroll
    for (int ks = 0; ks < L::KS_E; ++ks) {
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 bm = *(const f32x4*)(b16 + L::P_BM + unit_of_h(ks, 0, j) * 4);       // unit_of_h(ks, h, j) = unit_of_h(ks, 0, j) + 4 h
            const float arg = fmaf(pf[2], bm.z, fmaf(pf[1], bm.y, pf[0] * bm.x));
            e[j] = adfp_sinf(arg);
        }
        split8<false>(e, eh[ks], el[ks], amax);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc;
    f16x8 hh[2], hl[2];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        bias_init(acc, bh + L::P_BP(i), 0);
        if (i == 0) mfma_chain_h<L::KS_E>(acc, wl + L::P_WP(0), 0, eh, el);
        else if (i == 3) {
            mfma_chain_h<L::KS_E>(acc, wl + L::P_WP(3), 0, eh, el);
            mfma_chain_h<2>(acc, wl + L::P_WP(3) + L::KS_E * 512, 0, hh, hl);
        } else mfma_chain_h<2>(acc, wl + L::P_WP(i), 0, hh, hl);
        relu_bias(acc, bh + L::P_BC(i), 0);
        mfma_chain_h<L::KS_C>(acc, wl + L::P_WC(i), 0, ch, cl);
        if (i < 4) {
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = acc[r];
            split8(t, hh[0], hl[0], amax);
            split8(t + 8, hh[1], hl[1], amax);
        }
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        const float* wo = bw + L::P_WO + o * 16;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s = fmaf(acc[r], wo[r], s);
        s += __shfl_xor(s, 32);
        out[o] = s + ((const float*)ldsu)[L::P_BO + o];
    }
}

struct DecodeLCArgs {
    PtsDev P; NormDev nb; double b[6];
    GridDev g_low, g_color;
    const unsigned* packed_low; const unsigned* packed_color;     // H images
    const unsigned char* flags;                                   // ADFP_F_BAND per point (or NULL)
    float* raw; float* w;
    int write_w, apply_bound;
    int* status; int* call_flag;
    int* pool;                 // k_decode_lc16: device counter of the chip-wide tile tail (claim_tile_pool), zero at launch, or NULL
};
template <int NT>
__global__ __launch_bounds__(NT, NT / 256) void k_decode_lc(DecodeLCArgs a) {
    using LL = DecLayoutH<32, 1>;
    using LC = DecLayoutH<32, 4>;
    __shared__ __attribute__((aligned(16))) unsigned lds_all[LL::P_TOTAL + LC::P_TOTAL];      // the low image, then the colour image
    __shared__ int s_next;
    unsigned* lds_low = lds_all;
    unsigned* lds_col = lds_all + LL::P_TOTAL;
    image_to_lds<NT, LL::P_TOTAL / 4>(lds_low, a.packed_low);
    image_to_lds<NT, LC::P_TOTAL / 4>(lds_col, a.packed_color);
    if (threadIdx.x == 0) s_next = NT / 64;
    __syncthreads();
    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
    const int lane_off = h * 128 + p * 4;
    const int count = a.P.n;
    const int ntiles = (count + 31) >> 5;
    float amax_low = image_out_of_range<LL::P_FLAG, LL::NFLAG>(lds_low) ? INFINITY : 0.f;
    float amax_col = image_out_of_range<LC::P_FLAG, LC::NFLAG>(lds_col) ? INFINITY : 0.f;
    // The second image lies beyond the 64 KB reach of a ds_read's immediate offset.  With its address a compile-time constant the
    // compiler materialised one address register per distinct offset and hoisted them all out of the tile loop (109 spilled
    // VGPRs); with the image's word offset an opaque register value every access is (offset + lane term) + small immediate again.
    for (int j = threadIdx.x >> 6, tile; (tile = claim_tile<NT / 64>(j, &s_next, ntiles)) >= 0;) {
        const int idx = tile * 32 + p;
        const bool valid = idx < count;
        const int q = valid ? idx : 0;
        float pn[3], pf[3];
        bool pnan, keep_occ;                            // everything the f64 point is needed for, so that it dies before the networks
        {
            double pt[3];
            load_point(a.P, q, pt);
            normalize3(a.nb, pt, pn);
            pf[0] = (float)pt[0]; pf[1] = (float)pt[1]; pf[2] = (float)pt[2];   // p.float() decoder.py:189
            pnan = (pt[0] != pt[0]) | (pt[1] != pt[1]) | (pt[2] != pt[2]);
            const unsigned f = a.flags ? a.flags[q] : 0u;
            // in-band points keep the true low value for the HIGH pass; the attention pass overwrites them afterwards
            keep_occ = (f & ADFP_F_BAND) || in_bound(pt, a.b) || !a.apply_bound;           // Renderer.py:64
        }
        float occ[1], rgb[4];
        int off_low = 0, off_col = LL::P_TOTAL;        // word offsets of the two images in lds_all, opaque and per tile: nothing that
        asm volatile("" : "+v"(off_low), "+v"(off_col));   // derives from them is loop invariant (hoisted address registers were spilled)
        const unsigned* img_low = lds_all + off_low;
        const unsigned* img_col = lds_all + off_col;
        decode_net_h<1>(img_low, a.g_low, pn, pf, h, lane_off, amax_low, occ);
        // the colour network starts here, not earlier: without the opaque pass the optimiser hoists its trilinear set-up and
        // gather above the low network's layers and the two networks' operand sets no longer fit 168 registers
        asm volatile("" : "+v"(pn[0]), "+v"(pn[1]), "+v"(pn[2]), "+v"(pf[0]), "+v"(pf[1]), "+v"(pf[2]), "+v"(occ[0]));
        __builtin_amdgcn_sched_barrier(0);
        decode_net_h<4>(img_col, a.g_color, pn, pf, h, lane_off, amax_col, rgb);
        if (valid && h == 0) {
            const float nanv = __builtin_nanf("");     // a NaN position renders NaN like the reference's (nan_point_outputs)
            const float o = pnan ? nanv : (keep_occ ? occ[0] : 100.f);
            *(f32x4*)(a.raw + 4ll * q) = pnan ? f32x4{nanv, nanv, nanv, o} : f32x4{rgb[0], rgb[1], rgb[2], o};
            if (a.write_w) a.w[q] = 1.f;
        }
    }
    report_range(a.status, amax_low, ADFP_STATUS_F16_RANGE_LOW, a.call_flag);
    report_range(a.status, amax_col, ADFP_STATUS_F16_RANGE_COLOR, a.call_flag);
}

// =============================================================================================
// attention fusion mlp_tsdf (a11) on f16 MFMA with the same 3-product split
// =============================================================================================
struct AttLayoutH {
    using F = AttLayout;
    // words
    static constexpr int P_A0 = 0;                               // [64][4] f32 = (w0, w1, b, 0), unit order
    static constexpr int P_W1 = 256;                             // 4 out-blocks x 4 k-steps
    static constexpr int P_B1 = P_W1 + 4 * 4 * 512;
    static constexpr int P_W2 = P_B1 + 128;                      // 4 out-blocks x 8 k-steps
    static constexpr int P_B2 = P_W2 + 4 * 8 * 512;
    static constexpr int P_W3 = P_B2 + 128;                      // 2 out-blocks x 8 k-steps
    static constexpr int P_B3 = P_W3 + 2 * 8 * 512;
    static constexpr int P_WO = P_B3 + 64;                       // [2 h][2 o][32] f32
    static constexpr int P_BO = P_WO + 128;
    static constexpr int P_FLAG = P_BO + 4;                       // range flags, one word per pack block (see DecLayoutH)
    static constexpr int NFLAG = (((P_FLAG + 511) / 256) + 3) & ~3;
    static constexpr int P_TOTAL = P_FLAG + NFLAG;
    static_assert((P_TOTAL + 255) / 256 <= NFLAG, "one flag word per pack block");
};

__device__ HSrc att_h_src(int t) {
    using L = AttLayoutH;
    using F = AttLayout;
    if (t < L::P_W1) {
        const int k = t >> 2, c = t & 3;
        return HSrc{0, c < 2 ? F::F_W0 + k * 2 + c : (c == 2 ? F::F_B0 + k : -1), -1};
    }
    auto chain = [](int u, int nks, int base, int ld) {
        const int blk = u / (nks * 512), v = u % (nks * 512);
        const int ks = v >> 9, part = (v >> 8) & 1, h = (v >> 7) & 1, row = (v >> 2) & 31, jp = (v & 3) * 2;
        return HSrc{1 + part, base + (blk * 32 + row) * ld + unit_of_h(ks, h, jp), base + (blk * 32 + row) * ld + unit_of_h(ks, h, jp + 1)};
    };
    if (t < L::P_B1) return chain(t - L::P_W1, 4, F::F_W1, 64);
    if (t < L::P_W2) return HSrc{0, F::F_B1 + (t - L::P_B1), -1};
    if (t < L::P_B2) return chain(t - L::P_W2, 8, F::F_W2, 128);
    if (t < L::P_W3) return HSrc{0, F::F_B2 + (t - L::P_B2), -1};
    if (t < L::P_B3) return chain(t - L::P_W3, 8, F::F_W3, 128);
    if (t < L::P_WO) return HSrc{0, F::F_B3 + (t - L::P_B3), -1};
    if (t < L::P_BO) {                                            // entry j of lane-half h <-> unit_of(j, h)
        const int u = t - L::P_WO, h = u >> 6, o = (u >> 5) & 1, j = u & 31;
        return HSrc{0, F::F_WO + o * 64 + unit_of(j, h), -1};
    }
    const int o = t - L::P_BO;
    return HSrc{0, o < 2 ? F::F_BO + o : -1, -1};
}
ADFP_DEV void pack_attention_h_block(int blk, const float* __restrict__ flat, unsigned* __restrict__ packed, int* __restrict__ status) {
    using L = AttLayoutH;
    const int t = blk * 256 + (int)threadIdx.x;
    HSrc s{0, -1, -1};
    float a = 0.f, b = 0.f;
    if (t < L::P_FLAG) {
        s = att_h_src(t);
        a = s.s0 < 0 ? 0.f : flat[s.s0]; b = s.s1 < 0 ? 0.f : flat[s.s1];
    }
    pack_block_flag(!(fmaxf(fabsf(a), fabsf(b)) < 65504.0f), packed + L::P_FLAG + blk, status, ADFP_STATUS_F16_RANGE_ATT);
    if (t >= L::P_FLAG) return;
    if (s.kind == 0) { packed[t] = s.s0 < 0 ? 0u : __float_as_uint(a); return; }
    a = f16_clamp(a); b = f16_clamp(b);          // out of range (flagged above / by pack_range_flag): stay finite, 0 x inf must not appear downstream
    const float ah = f16_hi_part(a), bh = f16_hi_part(b);
    _Float16 x, y;
    if (s.kind == 1) { x = (_Float16)ah; y = (_Float16)bh; }
    else { x = (_Float16)(a - ah); y = (_Float16)(b - bh); }
    packed[t] = (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__global__ void k_pack_attention_h(const float* __restrict__ flat, unsigned* __restrict__ packed, int* __restrict__ status) { pack_attention_h_block((int)blockIdx.x, flat, packed, status); }

// relu(v) with its activity bit shifted into `m` (see relu_bias_mask)
ADFP_DEV float relu_mask(float v, unsigned& m) {
    const float t = relu_f(v);
    m = __builtin_amdgcn_alignbit(m, 0u - __float_as_uint(t), 31);
    return t;
}
// TRAIN: besides its outputs the kernel leaves what the f16-split attention backward needs (k_attention_bwd_h): per lane half
// 7 words -- the ReLU masks of the four layers (word 0: layer 0, value 8 ks + j at bit 31 - (8 ks + j); words 1-2 / 3-4: layers
// 1 / 2, out-block ob in word ob >> 1, its register r at bit 31 - 16 (ob & 1) - r; word 5: layer 3 likewise) and the softmax
// weight (half 0: a0, half 1: a1) -- and, when a.act is set, the layer inputs (AttStage columns [0, 416)).
#define ADFP_ATT_MASK_WORDS 14
// NT = 512 (two waves per SIMD, 256 registers each) for inference; the TRAIN variant needs ~330 registers (masks, the staged layer
// inputs held until their stores issue) and spilled 79 of them at 512 threads -- it runs at NT = 256, one wave per SIMD with the
// whole register file, on the 54 000 in-band rows of an iteration (1.7 tiles per wave either way).
template <int TRAIN, int NT = 512>
__global__ __launch_bounds__(NT) void k_attention_h(AttArgs a) {
    using A = AttLayoutH;
    using ST = AttStage;
    __shared__ __attribute__((aligned(16))) unsigned ldsu[A::P_TOTAL];
    __shared__ int s_next;
    image_to_lds<NT, A::P_TOTAL / 4>(ldsu, a.packed);
    if (threadIdx.x == 0) s_next = NT / 64;
    __syncthreads();
    const float* lds = (const float*)ldsu;
    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
    const int lane_off = h * 128 + p * 4;
    const int count = a.count_ptr ? *a.count_ptr : a.n_rows;
    const int ntiles = (count + 31) >> 5;
    float amax = image_out_of_range<A::P_FLAG, A::NFLAG>(ldsu) ? INFINITY : 0.f;
    for (int j = threadIdx.x >> 6, tile; (tile = claim_tile<NT / 64>(j, &s_next, ntiles)) >= 0;) {
        const int idx = tile * 32 + p;
        const bool valid = idx < count;
        const int ii = valid ? idx : 0;
        const float occ = a.att_occ[ii], u = a.att_u[ii];
        unsigned mk[6] = {0u, 0u, 0u, 0u, 0u, 0u};
        float* srow = (TRAIN && a.act && valid) ? a.act + (long long)idx * 416 : nullptr;
        if constexpr (TRAIN) if (srow) stage_head(srow, ST::AX, h, f32x4{occ, u, 1.f, 0.f});
        // layer 0 (2 -> 64) on the VALU
        f16x8 xh[8], xl[8];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float t8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 t = *(const f32x4*)(lds + A::P_A0 + unit_of_h(ks, h, j) * 4);
                const float v = fmaf(u, t.y, fmaf(occ, t.x, t.z));
                t8[j] = TRAIN ? relu_mask(v, mk[0]) : relu_f(v);
            }
            if constexpr (TRAIN) if (srow) {
                *(f32x4*)(srow + ST::AH0 + 32 * (ks >> 1) + 16 * (ks & 1) + 4 * h) = f32x4{t8[0], t8[1], t8[2], t8[3]};
                *(f32x4*)(srow + ST::AH0 + 32 * (ks >> 1) + 16 * (ks & 1) + 8 + 4 * h) = f32x4{t8[4], t8[5], t8[6], t8[7]};
            }
            split8(t8, xh[ks], xl[ks], amax);
        }
        __builtin_amdgcn_sched_barrier(0);
        // layer 1: 64 -> 128
        f16x8 yh[8], yl[8];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            f32x16 acc;
            bias_init(acc, lds + A::P_B1 + 32 * ob, h);
            mfma_chain_h<4>(acc, ldsu + A::P_W1 + ob * 4 * 512, lane_off, xh, xl);
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = TRAIN ? relu_mask(acc[r], mk[1 + (ob >> 1)]) : relu_f(acc[r]);
            if constexpr (TRAIN) if (srow) stage_block(srow, ST::AH1 + 32 * ob, h, t, 0);
            split8(t, yh[2 * ob], yl[2 * ob], amax);
            split8(t + 8, yh[2 * ob + 1], yl[2 * ob + 1], amax);
        }
        // layer 2: 128 -> 128
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            f32x16 acc;
            bias_init(acc, lds + A::P_B2 + 32 * ob, h);
            mfma_chain_h<8>(acc, ldsu + A::P_W2 + ob * 8 * 512, lane_off, yh, yl);
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = TRAIN ? relu_mask(acc[r], mk[3 + (ob >> 1)]) : relu_f(acc[r]);
            if constexpr (TRAIN) if (srow) stage_block(srow, ST::AH2 + 32 * ob, h, t, 0);
            split8(t, xh[2 * ob], xl[2 * ob], amax);
            split8(t + 8, xh[2 * ob + 1], xl[2 * ob + 1], amax);
        }
        // layer 3: 128 -> 64, output 64 -> 2 on the VALU in f32
        float l0 = 0.f, l1 = 0.f;
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            f32x16 acc;
            bias_init(acc, lds + A::P_B3 + 32 * ob, h);
            mfma_chain_h<8>(acc, ldsu + A::P_W3 + ob * 8 * 512, lane_off, xh, xl);
            const float* w0 = lds + A::P_WO + (h * 2 + 0) * 32 + 16 * ob;
            const float* w1 = lds + A::P_WO + (h * 2 + 1) * 32 + 16 * ob;
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = TRAIN ? relu_mask(acc[r], mk[5]) : relu_f(acc[r]);
                t[r] = v;
                l0 = fmaf(v, w0[r], l0);
                l1 = fmaf(v, w1[r], l1);
            }
            if constexpr (TRAIN) if (srow) stage_block(srow, ST::AH3 + 32 * ob, h, t, 0);
        }
        l0 += __shfl_xor(l0, 32); l1 += __shfl_xor(l1, 32);
        l0 += lds[A::P_BO]; l1 += lds[A::P_BO + 1];
        // softmax over 2, convex blend (decoder.py:255-258)
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        const float den = e0 + e1;
        const float a0 = e0 / den, a1 = e1 / den;
        const float fused = a0 * occ + a1 * u;
        if constexpr (TRAIN) if (valid) {
            unsigned* mrow = a.masks + ((long long)idx * 2 + h) * (ADFP_ATT_MASK_WORDS / 2);
#pragma unroll
            for (int k = 0; k < 6; ++k) mrow[k] = mk[k];
            mrow[6] = __float_as_uint(h ? a1 : a0);
        }
        if (valid && h == 0) {
            const int q = a.list ? a.list[ii] : ii;              // list == NULL: mlp_tsdf.forward on explicit (occ, u) rows
            const bool inb = !a.flags || (a.flags[q] & ADFP_F_INBOUND) != 0;
            a.raw[4ll * q + 3] = (inb || !a.apply_bound) ? fused : 100.f;   // Renderer.py:64
            a.w[q] = a1;
        }
    }
    report_range(a.status, amax, ADFP_STATUS_F16_RANGE_ATT, a.call_flag);
}
