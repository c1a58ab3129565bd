// adfp_decode_g.h -- LOW + COLOR decoder in one launch (stage color, inference) on v_mfma_f32_16x16x32_f16.
//
// Same arithmetic as k_decode_lc (adfp_decode_h.h): every f32 operand split a = a_hi + a_lo into two f16, three MFMA products
// per f32 product, f32 accumulation; same per-point work (f64 point, f64 normalisation, 8-corner gather, Fourier features with
// exact turn reduction, 5 layers, VALU output layer).  What changes is the MFMA SHAPE: the 32 units x 32 points of a layer are
// four 16 x 16 output blocks (2 out-blocks x 2 point-blocks), each fed K = 32 inputs per instruction, instead of one 32 x 32 block
// fed K = 16.  Twice the instructions at half the pipe cycles each -- the same wave cycles per tile (43 200 against 43 000 in the
// decoder-shaped loop of tools/micro/mfma_shape_ab.hip) -- but the chip HOLDS A HIGHER CLOCK on this shape under load (2.21 against
// 2.09 GHz there; MI355X_MICROARCH.md "DVFS give-back" (7)): wall time -6.5 % in the microbenchmark.
//
// Lane roles.  lane l = (n = l & 15, g = l >> 4).  A tile is 32 consecutive points = two point-blocks of 16; lane (n, g) serves
// point n of BOTH blocks with the 8 inputs / 8 outputs of K-group g:
//   B operand of point-block pb: the lane's 8 values of point 16 pb + n, K-slot j <-> input unit16(8 g + j)
//   D of (out-block ob, point-block pb): rows 4 g + r of the block = output units 16 ob + 4 g + r of point 16 pb + n
// with unit16(k) = 16 ((k & 7) >> 2) + 4 (k >> 3) + (k & 3): the two D registers-quads of a point-block (ob = 0, 1) ARE the next
// layer's B operand of that block -- as with the 32 x 32 shape there is no lane movement and no LDS round trip between layers.
//
// Front end.  The per-point work that does not factor over K-groups (f64 point, normalisation, trilinear set-up, the gather's
// address arithmetic) is done ONCE per point like before: lanes 0-31 take the points of block 0, lanes 32-63 those of block 1
// ("front point" 16 (g >> 1) + n, front half g & 1 = which 16 of the 32 channels the lane gathers, gather16's own split), and the
// halves hand each other the eight values the other needs with v_permlane32_swap_b32 (one instruction per register pair, no
// LDS): 8 swaps for the grid features, 3 for the position.  The Fourier features need no exchange: a lane computes its 24
// features of BOTH its points from one read of each Fourier row (half the LDS reads of the 32 x 32 kernel).
#pragma once
#include "adfp_decode_h.h"

__host__ __device__ constexpr int unit16(int k) { return 16 * ((k & 7) >> 2) + 4 * (k >> 3) + (k & 3); }

// "G" image of one 32-channel decoder (words).  A K = 32 group of a [32 out x K in] block is 1024 words:
// [out-block 0 | 1][hi | lo][g = 0..3][row i = 0..15][8 halves]: lane (n = i, g) reads its A operand (hi, lo) of an out-block with
// two ds_read_b128 at word 4 l -- 64 lanes, 64 consecutive 16-byte pieces, conflict-free.  The same 4 reads per 32 inputs as the H
// image's two k-steps.
template <int CDIM, int NOUT>
struct DecLayoutG {
    using F = DecLayout<CDIM, NOUT>;
    static constexpr int KG_E = 3;                                    // K-groups of the 96 (93) Fourier features
    static constexpr int KG_C = CDIM / 32;                            // K-groups of the grid features (high decoder: own grid, then the low grid)
    __host__ __device__ static constexpr int kg(int i) { return i == 0 ? KG_E : (i == 3 ? KG_E + 1 : 1); }
    static constexpr int P_BM = 0;                                    // [96 rows in K order][4] f32 = (bx, by, bz, 0)
    __host__ __device__ static constexpr int layer_words(int i) { return kg(i) * 1024 + 32 + KG_C * 1024 + 32; }
    __host__ __device__ static constexpr int P_WP(int i) {
        int o = 384;
        for (int k = 0; k < i; ++k) o += layer_words(k);
        return o;
    }
    __host__ __device__ static constexpr int P_BP(int i) { return P_WP(i) + kg(i) * 1024; }     // [32] f32, unit order
    __host__ __device__ static constexpr int P_WC(int i) { return P_BP(i) + 32; }
    __host__ __device__ static constexpr int P_BC(int i) { return P_WC(i) + KG_C * 1024; }
    static constexpr int P_WO = P_WP(5);                               // [NOUT][32] f32, unit order
    static constexpr int P_BO = P_WO + NOUT * 32;
    static constexpr int P_FLAG = P_BO + 4;                            // range flags, one word per pack block (see DecLayoutH)
    static constexpr int NFLAG = (((P_FLAG + 511) / 256) + 3) & ~3;
    static constexpr int P_TOTAL = P_FLAG + NFLAG;
    static_assert((P_TOTAL + 255) / 256 <= NFLAG, "one flag word per pack block");
};

template <int CDIM, int NOUT>
__device__ HSrc dec_g_src(int t) {
    using L = DecLayoutG<CDIM, NOUT>;
    using F = DecLayout<CDIM, NOUT>;
    if (t < 384) {
        const int row = t >> 2, c = t & 3;                            // row = 32 kg + k: feature 32 kg + unit16(k)
        const int f = (row & ~31) + unit16(row & 31);
        return HSrc{0, (f < 93 && c < 3) ? F::F_EB + c * 93 + f : -1, -1};
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        if (t < L::P_BP(i) || (t >= L::P_WC(i) && t < L::P_BC(i))) {
            const bool fc = t >= L::P_WC(i);
            const int u = t - (fc ? L::P_WC(i) : L::P_WP(i));
            const int grp = u >> 10, ob = (u >> 9) & 1, part = (u >> 8) & 1, g = (u >> 6) & 3, row = 16 * ob + ((u >> 2) & 15), jp = (u & 3) * 2;
            int src[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int col = 32 * grp + unit16(8 * g + jp + e);            // input unit in the layer's own numbering: [features | hidden]
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
        return HSrc{0, F::F_OW + u, -1};                              // [NOUT][32], the flat order
    }
    const int o = t - L::P_BO;
    return HSrc{0, o < NOUT ? F::F_OB + o : -1, -1};
}

template <int CDIM, int NOUT>
ADFP_DEV void pack_decoder_g_block(int blk, const float* __restrict__ flat, unsigned* __restrict__ packed, int* __restrict__ status, int bit) {
    using L = DecLayoutG<CDIM, NOUT>;
    const int t = blk * 256 + (int)threadIdx.x;
    HSrc s{0, -1, -1};
    float a = 0.f, b = 0.f;
    if (t < L::P_FLAG) {
        s = dec_g_src<CDIM, NOUT>(t);
        a = s.s0 < 0 ? 0.f : flat[s.s0]; b = s.s1 < 0 ? 0.f : flat[s.s1];
    }
    pack_block_flag(!(fmaxf(fabsf(a), fabsf(b)) < 65504.0f), packed + L::P_FLAG + blk, status, bit);
    if (t >= L::P_FLAG) return;
    if (s.kind == 0) { packed[t] = s.s0 < 0 ? 0u : __float_as_uint(a); return; }
    a = f16_clamp(a); b = f16_clamp(b);
    const float ah = f16_hi_part(a), bh = f16_hi_part(b);
    _Float16 x, y;
    if (s.kind == 1) { x = (_Float16)ah; y = (_Float16)bh; }
    else { x = (_Float16)(a - ah); y = (_Float16)(b - bh); }
    packed[t] = (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
template <int CDIM, int NOUT>
__global__ void k_pack_decoder_g(const float* __restrict__ flat, unsigned* __restrict__ packed, int* __restrict__ status, int bit) { pack_decoder_g_block<CDIM, NOUT>((int)blockIdx.x, flat, packed, status, bit); }

typedef float f32x4g __attribute__((ext_vector_type(4)));

// debug build (-DADFP_STAMPS_G, tools/phase_g.sh): wave-cycles per phase of a tile, summed over waves.  Slots: [0..7] k_decode_high_g,
// [8..15] / [16..23] the low / colour network of k_decode_lc16; phase 0 = tile claim + point, 1 = gather + exchange + split,
// 2 = Fourier features, 3 = the five layers, 4 = output layer, 5 = stores
#ifdef ADFP_STAMPS_G
__device__ unsigned long long g_phase_g[24];
__device__ unsigned long long g_wave_span_g[2 * 4096];      // k_decode_lc16: per wave (wall-clock start, end), 100 MHz
#define ADFP_PHG_PARAMS , unsigned long long* ph_, unsigned long long& last_
#define ADFP_PHG_ARGS(base) , ph_ + (base), last_
#define ADFP_PHG(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = clock64(); \
                         __builtin_amdgcn_sched_barrier(0); ph_[k] += now_ - last_; last_ = now_; } while (0)
#else
#define ADFP_PHG_PARAMS
#define ADFP_PHG_ARGS(base)
#define ADFP_PHG(k) do {} while (0)
#endif

// x (first operand): lanes 32-63 receive y of lane l - 32;  y: lanes 0-31 receive x of lane l + 32  (tools/micro/layout_probe_16x16x32.hip)
ADFP_DEV void swap_halves(float& x, float& y) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r[0]); y = __uint_as_float(r[1]);
}

// NG K-groups of a chain: acc[ob][pb] += W[out-block ob][group] * x[group][pb], 3-product split
template <int NG>
ADFP_DEV void mfma_chain_g(f32x4g acc[2][2], const unsigned* __restrict__ w, const f16x8 (*xh)[2], const f16x8 (*xl)[2]) {
#pragma unroll
    for (int kg = 0; kg < NG; ++kg) {
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            const f16x8 ah = __builtin_bit_cast(f16x8, *(const u32x4*)(w + kg * 1024 + ob * 512));
            const f16x8 al = __builtin_bit_cast(f16x8, *(const u32x4*)(w + kg * 1024 + ob * 512 + 256));
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                acc[ob][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh[kg][pb], acc[ob][pb], 0, 0, 0);
                acc[ob][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl[kg][pb], acc[ob][pb], 0, 0, 0);
                acc[ob][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh[kg][pb], acc[ob][pb], 0, 0, 0);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting the next chain's LDS reads
}

// One network on a tile.  pn: the lane's FRONT point, normalised (gather); pf[pb]: the positions of the lane's two points (Fourier
// features); out[pb][o]: the network's outputs for point 16 pb + n, valid on every lane.
//
// TRAIN (the training forward of the fused low + colour launch): the network also leaves what the f16-split backward reads -- in
// the formats of k_decode_h<..., TRAIN> (adfp_decode_h.h), so that the backward kernels do not know which forward ran:
//   mw[k], k = 0..2: the ReLU-mask words of the lane's FRONT point 16 (g >> 1) + n, lane half g & 1 (adfp_train_state.masks_*:
//        [row][half][3] words, 16-bit field of layer i, bit 15 - r <-> unit kmapH(r, half)).  The lane holds units 16 ob + 4 g + r of
//        its two points = bits 15 - 8 ob - 4 (g >> 1) - r of the field of half g & 1: half of a field; the other half sits in lane
//        l ^ 32, and ONE v_permlane32_swap per word brings the two halves of point-block 0 together on the lower lanes and those
//        of point-block 1 on the upper lanes -- each lane ends up with the complete words of its front point.
//   srow0 / rowbits: the X piece of point 16 pb + n's staging row (DecStage: [head: not written] | c | h_0..h_4, natural unit
//        order).  The lane's units 16 ob + 4 g .. + 3 are one 16-byte piece: 2 pieces of c and of every h_i per point, 24 stores
//        per lane and tile.  The head [x, y, z, 1, 0 ...] (128 B of the 896) is NOT stored: k_decode_bwd_roles / _fused recompute
//        the Fourier features from the points, and k_outer_h builds the head from the points (OuterHArgs.x_skip4).
// the training rows' stores: plain, or (timing experiment -DADFP_EXP_NT_ROWS) non-temporal -- written once, read once ~300 us later
ADFP_DEV void row_store(float* p, const f32x4 v) {
#if defined(ADFP_EXP_NT_ROWS)
    __builtin_nontemporal_store(v, (f32x4*)p);
#else
    *(f32x4*)p = v;
#endif
}
template <int CDIM, int NOUT, int TRAIN = 0>
ADFP_DEV void decode_net_g(const unsigned* __restrict__ ldsu, const GridDev& grid, const GridDev& grid1, const float pn[3], const float (*pf)[3],
                           int lane, float& amax, float (*out)[NOUT], unsigned* __restrict__ mw, float* srow0, unsigned rowbits ADFP_PHG_PARAMS) {
    using L = DecLayoutG<CDIM, NOUT>;
    using ST = DecStage<CDIM>;
    const int g = lane >> 4;
    // TRAIN: srow0 = the row of point n (point-block 0), the row of point 16 + n lies 16 rows on; rowbits bit pb = that row exists
    float* srow[2] = {nullptr, nullptr};
    if constexpr (TRAIN) {
        srow[0] = (rowbits & 1u) ? srow0 : nullptr;
        srow[1] = (rowbits & 2u) ? srow0 + 16 * ST::NXM : nullptr;
#if defined(ADFP_EXP_TRAIN_NOX)            // timing experiment: no layer-input rows at all (masks only)
        srow[0] = nullptr; srow[1] = nullptr;
#endif
    }
    // every LDS access below is one of three lane-dependent bases plus an immediate: the weight rows (4 l), the unit-order rows
    // (biases, output layer: 4 g) and the Fourier rows (32 g)
    const unsigned* wl = ldsu + 4 * lane;
    const float* b4 = (const float*)ldsu + 4 * g;
    const float* brow = (const float*)ldsu + 32 * g;
    f16x8 ch[L::KG_C][2], cl[L::KG_C][2];
#pragma unroll
    for (int kc = 0; kc < L::KG_C; ++kc) {          // high decoder: K-group 0 = its own grid, K-group 1 = the low grid (decoder.py:182-187)
        float c[16];
#if defined(ADFP_EXP_NOGATHER)     // timing experiments (tools/ab_high.sh): no gather at all / no second gather
        for (int r = 0; r < 16; ++r) c[r] = pn[r % 3] * (float)(r + 1 + 16 * kc);
#elif defined(ADFP_EXP_NOGATHER1)
        if (kc == 0) gather16(grid, pn, g & 1, c); else for (int r = 0; r < 16; ++r) c[r] = pn[r % 3] * (float)(r + 1);
#else
        gather16(kc == 0 ? grid : grid1, pn, g & 1, c);   // c[r] <-> channel kmapH(r, g & 1) of the front point: pieces g&1, +2, +4, +6 of the voxel line
#endif
        // lower lanes (block 0) keep K-group g = c[0..3], c[8..11] and give K-group g + 2 = c[4..7], c[12..15]; upper lanes (block 1,
        // g = 2, 3) keep c[4..7], c[12..15] and give c[0..3], c[8..11]: swap_halves(x, y) moves x.upper <-> y.lower
        float x[8] = {c[0], c[1], c[2], c[3], c[8], c[9], c[10], c[11]};
        float y[8] = {c[4], c[5], c[6], c[7], c[12], c[13], c[14], c[15]};
#pragma unroll
        for (int s = 0; s < 8; ++s) swap_halves(x[s], y[s]);
#if !defined(ADFP_EXP_TRAIN_NOC)          // timing experiment (tools/build_ab_libs.sh): the training forward without the c piece of the rows
        if constexpr (TRAIN) {
            static_assert(!TRAIN || CDIM == 32, "training rows: 32-channel decoders");
            if (srow[0]) { row_store(srow[0] + ST::xm(ST::SC) + 4 * g, f32x4{x[0], x[1], x[2], x[3]}); row_store(srow[0] + ST::xm(ST::SC) + 16 + 4 * g, f32x4{x[4], x[5], x[6], x[7]}); }
            if (srow[1]) { row_store(srow[1] + ST::xm(ST::SC) + 4 * g, f32x4{y[0], y[1], y[2], y[3]}); row_store(srow[1] + ST::xm(ST::SC) + 16 + 4 * g, f32x4{y[4], y[5], y[6], y[7]}); }
        }
#endif
        split8(x, ch[kc][0], cl[kc][0], amax);      // block 0: channels unit16(8 g + j)
        split8(y, ch[kc][1], cl[kc][1], amax);      // block 1
    }
    ADFP_PHG(1);
    f16x8 eh[L::KG_E][2], el[L::KG_E][2];
#if defined(ADFP_PB_MFMA)
    // p @ B on the f32 matrix pipe (round 6; role P of k_decode_bwd_roles does the same, bit-identical to the fma chain below:
    // tools/micro/mfma_f32_fma_order.hip).  One v_mfma_f32_32x32x2_f32 pair per K-group: D[slot][point], K = 3 as (x, y) then (z, 0)
    // -- each k step an f32 fma, so an element is fmaf(z, bz, fmaf(y, by, x * bx)).  Operands: A = the Fourier rows, lane
    // (i = l & 31, k = l >> 5) reads ONE row -- the row of slot sigma(i) -- instead of eight; B = the positions, lane (c = l & 31)
    // = point 16 (g & 1) + n.  D: lane (n, g) holds, for ITS point pb = g & 1, the 16 rows (r & 3) + 8 (r >> 2) + 4 (g >> 1); sigma
    // places K-group 2 (g >> 1)'s eight slots in registers 0-7 and K-group 2 (g >> 1) + 1's in registers 8-15, so the lane pair
    // (l, l ^ 16) -- same rows, the two point blocks -- trade what the other needs with eight v_permlane16_swap: afterwards
    // a[s] = slot 8 g + s of point n, b[s] = the same slot of point 16 + n, on every lane.
    {
        const int i31 = lane & 31;
        const int slot = 16 * ((i31 >> 2) & 1) + 8 * (i31 >> 4) + (i31 & 3) + 4 * ((i31 >> 3) & 1);
        const float* bsl = (const float*)ldsu + L::P_BM + 4 * slot;
        const bool k1 = lane >= 32, odd = (g & 1) != 0;
        const float px = odd ? pf[1][0] : pf[0][0], py = odd ? pf[1][1] : pf[0][1], pz = odd ? pf[1][2] : pf[0][2];
        const float pxy = k1 ? py : px, pz0 = k1 ? 0.f : pz;
#pragma unroll
        for (int kg = 0; kg < L::KG_E; ++kg) {
            const f32x4 bm = *(const f32x4*)(bsl + 128 * kg);
            f32x16 av;
#pragma unroll
            for (int r = 0; r < 16; ++r) av[r] = 0.f;
            av = __builtin_amdgcn_mfma_f32_32x32x2f32(k1 ? bm.y : bm.x, pxy, av, 0, 0, 0);
            av = __builtin_amdgcn_mfma_f32_32x32x2f32(k1 ? 0.f : bm.z, pz0, av, 0, 0, 0);
            float e0[8], e1[8];
#pragma unroll
            for (int s_ = 0; s_ < 8; ++s_) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(av[s_]), __float_as_uint(av[8 + s_]), false, false);
                e0[s_] = adfp_sinf(__uint_as_float(sw[0]));
                e1[s_] = adfp_sinf(__uint_as_float(sw[1]));
            }
            split8<false>(e0, eh[kg][0], el[kg][0], amax);
            split8<false>(e1, eh[kg][1], el[kg][1], amax);
        }
    }
#else
#pragma unroll
    for (int kg = 0; kg < L::KG_E; ++kg) {
        float e0[8], e1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 bm = *(const f32x4*)(brow + L::P_BM + (32 * kg + j) * 4);            // row 32 kg + 8 g + j
            e0[j] = adfp_sinf(fmaf(pf[0][2], bm.z, fmaf(pf[0][1], bm.y, pf[0][0] * bm.x)));
            e1[j] = adfp_sinf(fmaf(pf[1][2], bm.z, fmaf(pf[1][1], bm.y, pf[1][0] * bm.x)));
        }
        split8<false>(e0, eh[kg][0], el[kg][0], amax);
        split8<false>(e1, eh[kg][1], el[kg][1], amax);
    }
#endif
    ADFP_PHG(2);
    __builtin_amdgcn_sched_barrier(0);
    f32x4g acc[2][2];
    f16x8 hh[1][2], hl[1][2];
    unsigned mk[3] = {0u, 0u, 0u};      // TRAIN: 16 bits per layer, two layers per register, pushed in the order ob, pb, r: [ob0 pb0 | ob0 pb1 | ob1 pb0 | ob1 pb1] nibbles
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) { const f32x4 t = *(const f32x4*)(b4 + L::P_BP(i) + 16 * ob); acc[ob][0] = t; acc[ob][1] = t; }
        if (i == 0) mfma_chain_g<L::KG_E>(acc, wl + L::P_WP(0), eh, el);
        else if (i == 3) {
            mfma_chain_g<L::KG_E>(acc, wl + L::P_WP(3), eh, el);
            mfma_chain_g<1>(acc, wl + L::P_WP(3) + L::KG_E * 1024, hh, hl);
        } else mfma_chain_g<1>(acc, wl + L::P_WP(i), hh, hl);
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {               // relu, then the fc_c bias (the chain below adds Wc c)
            const f32x4 t = *(const f32x4*)(b4 + L::P_BC(i) + 16 * ob);
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = relu_f(acc[ob][pb][r]);
                    if constexpr (TRAIN) mk[i >> 1] = __builtin_amdgcn_alignbit(mk[i >> 1], 0u - __float_as_uint(v), 31);   // relu_bias_mask's predicate
                    acc[ob][pb][r] = v + t[r];
                }
        }
        mfma_chain_g<L::KG_C>(acc, wl + L::P_WC(i), ch, cl);
        if constexpr (TRAIN) {
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) if (srow[pb]) {
                row_store(srow[pb] + ST::xm(ST::SH(i)) + 4 * g, f32x4{acc[0][pb][0], acc[0][pb][1], acc[0][pb][2], acc[0][pb][3]});
                row_store(srow[pb] + ST::xm(ST::SH(i)) + 16 + 4 * g, f32x4{acc[1][pb][0], acc[1][pb][1], acc[1][pb][2], acc[1][pb][3]});
            }
        }
        if (i < 4) {
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const float t[8] = {acc[0][pb][0], acc[0][pb][1], acc[0][pb][2], acc[0][pb][3],
                                    acc[1][pb][0], acc[1][pb][1], acc[1][pb][2], acc[1][pb][3]};
                split8(t, hh[0][pb], hl[0][pb], amax);
            }
        }
    }
    ADFP_PHG(3);
    // output_linear on the VALU in f32: the lane holds units 16 ob + 4 g + r of its two points; the four K-groups add up by two
    // exchanges (lanes l ^ 16, l ^ 32)
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        const f32x4 w0 = *(const f32x4*)(b4 + L::P_WO + 32 * o), w1 = *(const f32x4*)(b4 + L::P_WO + 32 * o + 16);
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) s = fmaf(acc[0][pb][r], w0[r], s);
#pragma unroll
            for (int r = 0; r < 4; ++r) s = fmaf(acc[1][pb][r], w1[r], s);
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            out[pb][o] = s + ((const float*)ldsu)[L::P_BO + o];
        }
    }
    ADFP_PHG(4);
    if constexpr (TRAIN) {
        // layer i's 16 bits: the low half of its register for i = 1, 3, 4, the high half for i = 0, 2; nibbles [ob0 pb0][ob0 pb1][ob1 pb0][ob1 pb1]
        const int sh = 4 * (g >> 1);
        unsigned wpb[2][3];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            unsigned f[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const unsigned m16 = (i == 4 || (i & 1)) ? (mk[i >> 1] & 0xFFFFu) : (mk[i >> 1] >> 16);
                const unsigned n0 = (m16 >> (12 - 4 * pb)) & 0xFu, n1 = (m16 >> (4 - 4 * pb)) & 0xFu;       // ob 0, ob 1 of this point-block
                f[i] = (n0 << (12 - sh)) | (n1 << (4 - sh));
            }
            wpb[pb][0] = f[0] | (f[1] << 16); wpb[pb][1] = f[2] | (f[3] << 16); wpb[pb][2] = f[4];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const auto r = __builtin_amdgcn_permlane32_swap(wpb[0][k], wpb[1][k], false, false);   // lower lanes: (own pb 0, partner's pb 0); upper: (partner's pb 1, own pb 1)
            mw[k] = r[0] | r[1];
        }
    }
}

// DecodeLCArgs is k_decode_lc's; packed_low / packed_color point at the G images
template <int NT>
__global__ __launch_bounds__(NT, NT / 256) void k_decode_lc16(DecodeLCArgs a) {
    using LL = DecLayoutG<32, 1>;
    using LC = DecLayoutG<32, 4>;
    __shared__ __attribute__((aligned(16))) unsigned lds_all[LL::P_TOTAL + LC::P_TOTAL];      // the low image, then the colour image
    __shared__ int s_next;
    __shared__ unsigned long long s_ring[ADFP_POOL_RING];
    unsigned* lds_low = lds_all;
    unsigned* lds_col = lds_all + LL::P_TOTAL;
    image_to_lds<NT, LL::P_TOTAL / 4>(lds_low, a.packed_low);
    image_to_lds<NT, LC::P_TOTAL / 4>(lds_col, a.packed_color);
    if (threadIdx.x == 0) s_next = NT / 64;
    if (threadIdx.x < ADFP_POOL_RING) s_ring[threadIdx.x] = 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int count = a.P.n;
    const int ntiles = (count + 31) >> 5;
    const TilePlan plan = tile_plan(ntiles, (int)gridDim.x, NT / 64, a.pool != nullptr);
    float amax_low = image_out_of_range<LL::P_FLAG, LL::NFLAG>(lds_low) ? INFINITY : 0.f;
    float amax_col = image_out_of_range<LC::P_FLAG, LC::NFLAG>(lds_col) ? INFINITY : 0.f;
#ifdef ADFP_STAMPS_G
    unsigned long long ph_[24] = {}, last_ = clock64();
    const unsigned long long wstart_ = wall_clock64();
#endif
    for (int j = threadIdx.x >> 6, tile; (tile = claim_tile_pool<NT / 64>(j, &s_next, s_ring, plan, ntiles, a.pool, a.status)) >= 0;) {
        // front point: point n of block 0 on lanes 0-31 (both K-groups g = 0, 1 of the pair), of block 1 on lanes 32-63
        const int idx = tile * 32 + 16 * (g >> 1) + n;
        const bool valid = idx < count;
        const int q = valid ? idx : 0;
        float pn[3], pf[2][3];
        bool pnan, keep_occ;                            // of the FRONT point: its lane with g & 1 == 0 stores the row
        {
            double pt[3];
            load_point(a.P, q, pt);
            normalize3(a.nb, pt, pn);
            const float f0 = (float)pt[0], f1 = (float)pt[1], f2 = (float)pt[2];   // p.float() decoder.py:189
            pnan = (pt[0] != pt[0]) | (pt[1] != pt[1]) | (pt[2] != pt[2]);
            const unsigned f = a.flags ? a.flags[q] : 0u;
            // in-band points keep the true low value for the HIGH pass; the attention pass overwrites them afterwards
            keep_occ = (f & ADFP_F_BAND) || in_bound(pt, a.b) || !a.apply_bound;           // Renderer.py:64
            pf[0][0] = f0; pf[0][1] = f1; pf[0][2] = f2; pf[1][0] = f0; pf[1][1] = f1; pf[1][2] = f2;
        }
        // both halves end up with block 0's position in pf[0] and block 1's in pf[1]
#pragma unroll
        for (int k = 0; k < 3; ++k) swap_halves(pf[0][k], pf[1][k]);
        float occ[2][1], rgb[2][4];
        int off_low = 0, off_col = LL::P_TOTAL;        // word offsets of the two images, opaque and per tile (see k_decode_lc)
        asm volatile("" : "+v"(off_low), "+v"(off_col));
#ifdef ADFP_STAMPS_G
        ph_[14] += 1;
#endif
        ADFP_PHG(8);
        decode_net_g<32, 1>(lds_all + off_low, a.g_low, a.g_low, pn, pf, lane, amax_low, occ, nullptr, nullptr, 0u ADFP_PHG_ARGS(8));
        asm volatile("" : "+v"(pn[0]), "+v"(pn[1]), "+v"(pn[2]), "+v"(occ[0][0]), "+v"(occ[1][0]));
        __builtin_amdgcn_sched_barrier(0);
        decode_net_g<32, 4>(lds_all + off_col, a.g_color, a.g_color, pn, pf, lane, amax_col, rgb, nullptr, nullptr, 0u ADFP_PHG_ARGS(16));
        // the front point's lane with g & 1 == 0 stores its row: lanes 0-15 block 0, lanes 32-47 block 1
        if (valid && (g & 1) == 0) {
            const int pb = g >> 1;
            const float nanv = __builtin_nanf("");     // a NaN position renders NaN like the reference's (nan_point_outputs)
            const float o = pnan ? nanv : (keep_occ ? (pb ? occ[1][0] : occ[0][0]) : 100.f);
            const f32x4 c4 = pb ? f32x4{rgb[1][0], rgb[1][1], rgb[1][2], o} : f32x4{rgb[0][0], rgb[0][1], rgb[0][2], o};
            *(f32x4*)(a.raw + 4ll * q) = pnan ? f32x4{nanv, nanv, nanv, o} : c4;
            if (a.write_w) a.w[q] = 1.f;
        }
        ADFP_PHG(21);
    }
#ifdef ADFP_STAMPS_G
    { const unsigned long long wend_ = wall_clock64();       // before the phase atomics below (3 072 waves x 16 atomics on 16 addresses)
      const int wv_ = blockIdx.x * (NT / 64) + (threadIdx.x >> 6); if (lane == 0 && wv_ < 4096) { g_wave_span_g[2 * wv_] = wstart_ | (ph_[14] << 48); g_wave_span_g[2 * wv_ + 1] = wend_; } }
#if ADFP_STAMPS_G != 2      // (= 2: spans only -- the 49 000 atomics below queue up in L2 in front of the still running waves' gathers and fake a tail)
    if (lane == 0) for (int k = 8; k < 24; ++k) atomicAdd(&g_phase_g[k], ph_[k]);
#endif
#endif
    report_range(a.status, amax_low, ADFP_STATUS_F16_RANGE_LOW, a.call_flag);
    report_range(a.status, amax_col, ADFP_STATUS_F16_RANGE_COLOR, a.call_flag);
}

// =============================================================================================
// The TRAINING forward of the fused launch (stage color, both networks f16-split, adfp_train_state with masks_low and masks_color):
// k_decode_lc16 plus the ReLU masks of both networks and, for a network whose weight gradients are wanted (act_* != NULL), its
// layer inputs.  A NaN position (a ray the Mapper's pre-filter drops) is decoded at the origin, as in k_decode_h<..., TRAIN>.
// Replaces k_decode_h<32, 1, LOW, 512, TRAIN> + k_decode_h<32, 4, COLOR, 512, TRAIN> of the training iteration.
// =============================================================================================
struct DecodeLCTrainArgs {
    DecodeLCArgs f;
    unsigned* masks_low; unsigned* masks_color;
    float* act_low; float* act_color;              // or NULL
    // SMALL batches (fewer tiles than half the launch's waves: a Tracker iteration, a 1 000-ray Mapper batch): a wave takes ONE network
    // of a tile instead of both in turn -- the launch is then the latency of one network's five dependent layers, not of two
    // (k_decode_lc16_train<NT, true>; the host sets the flag and picks the instantiation)
    int split_networks;
};
template <int NT, bool SPLIT = false>
__global__ __launch_bounds__(NT, NT / 256) void k_decode_lc16_train(DecodeLCTrainArgs t) {
    const DecodeLCArgs& a = t.f;
    using LL = DecLayoutG<32, 1>;
    using LC = DecLayoutG<32, 4>;
    using ST = DecStage<32>;
    __shared__ __attribute__((aligned(16))) unsigned lds_all[LL::P_TOTAL + LC::P_TOTAL];
    __shared__ int s_next;
    unsigned* lds_low = lds_all;
    unsigned* lds_col = lds_all + LL::P_TOTAL;
    image_to_lds<NT, LL::P_TOTAL / 4>(lds_low, a.packed_low);
    image_to_lds<NT, LC::P_TOTAL / 4>(lds_col, a.packed_color);
    if (threadIdx.x == 0) s_next = NT / 64;
    __syncthreads();
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int count = a.P.n;
    const int ntiles = (count + 31) >> 5;
    float amax_low = image_out_of_range<LL::P_FLAG, LL::NFLAG>(lds_low) ? INFINITY : 0.f;
    float amax_col = image_out_of_range<LC::P_FLAG, LC::NFLAG>(lds_col) ? INFINITY : 0.f;
    constexpr bool split = SPLIT;                   // a template parameter: the whole-batch kernel carries none of the branches below
    const int njobs = split ? 2 * ntiles : ntiles;
    for (int j = threadIdx.x >> 6, job; (job = claim_tile<NT / 64>(j, &s_next, njobs)) >= 0;) {
        const int tile = split ? job >> 1 : job;
        const bool do_low = !split || (job & 1) == 0, do_col = !split || (job & 1) == 1;      // wave-uniform
        const int idx = tile * 32 + 16 * (g >> 1) + n;
        const bool valid = idx < count;
        const int q = valid ? idx : 0;
        float pn[3], pf[2][3];
        bool pnan, keep_occ;
        {
            double pt[3];
            load_point(a.P, q, pt);
            normalize3(a.nb, pt, pn);
            pnan = (pt[0] != pt[0]) | (pt[1] != pt[1]) | (pt[2] != pt[2]);
            const float f0 = pnan ? 0.f : (float)pt[0], f1 = pnan ? 0.f : (float)pt[1], f2 = pnan ? 0.f : (float)pt[2];
            const unsigned f = a.flags ? a.flags[q] : 0u;
            keep_occ = (f & ADFP_F_BAND) || in_bound(pt, a.b) || !a.apply_bound;
            pf[0][0] = f0; pf[0][1] = f1; pf[0][2] = f2; pf[1][0] = f0; pf[1][1] = f1; pf[1][2] = f2;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) swap_halves(pf[0][k], pf[1][k]);
        // the rows of the lane's two points (point 16 pb + n of the tile): row of point-block 0 + which of the two exist
        const int i0 = tile * 32 + n;
        const unsigned rowbits = (i0 < count ? 1u : 0u) | (i0 + 16 < count ? 2u : 0u);
#ifdef ADFP_STAMPS_G
        unsigned long long ph_[24] = {}, last_ = 0;      // not collected for the training kernel
#endif
        float occ[2][1], rgb[2][4];
        unsigned mw[3];
        int off_low = 0, off_col = LL::P_TOTAL;
        asm volatile("" : "+v"(off_low), "+v"(off_col));
        if constexpr (split) {
            occ[0][0] = 0.f; occ[1][0] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { rgb[0][k] = 0.f; rgb[1][k] = 0.f; }
        }
        if (do_low) {
            decode_net_g<32, 1, 1>(lds_all + off_low, a.g_low, a.g_low, pn, pf, lane, amax_low, occ, mw, t.act_low + (long long)i0 * ST::NXM, t.act_low ? rowbits : 0u ADFP_PHG_ARGS(8));
            if (valid) { unsigned* mrow = t.masks_low + ((long long)q * 2 + (g & 1)) * 3; mrow[0] = mw[0]; mrow[1] = mw[1]; mrow[2] = mw[2]; }
        }
        asm volatile("" : "+v"(pn[0]), "+v"(pn[1]), "+v"(pn[2]), "+v"(occ[0][0]), "+v"(occ[1][0]));
        __builtin_amdgcn_sched_barrier(0);
        if (do_col) {
            decode_net_g<32, 4, 1>(lds_all + off_col, a.g_color, a.g_color, pn, pf, lane, amax_col, rgb, mw, t.act_color + (long long)i0 * ST::NXM, t.act_color ? rowbits : 0u ADFP_PHG_ARGS(16));
            if (valid) { unsigned* mrow = t.masks_color + ((long long)q * 2 + (g & 1)) * 3; mrow[0] = mw[0]; mrow[1] = mw[1]; mrow[2] = mw[2]; }
        }
        if (valid && (g & 1) == 0) {
            const int pb = g >> 1;
            const float nanv = __builtin_nanf("");
            const float o = pnan ? nanv : (keep_occ ? (pb ? occ[1][0] : occ[0][0]) : 100.f);
            const f32x4 c4 = pb ? f32x4{rgb[1][0], rgb[1][1], rgb[1][2], o} : f32x4{rgb[0][0], rgb[0][1], rgb[0][2], o};
            const f32x4 row = pnan ? f32x4{nanv, nanv, nanv, o} : c4;
            if (!split) {
                *(f32x4*)(a.raw + 4ll * q) = row;
                if (a.write_w) a.w[q] = 1.f;
            } else if (do_low) {                     // the row's two parts leave from the two waves that made them
                a.raw[4ll * q + 3] = row.w;
                if (a.write_w) a.w[q] = 1.f;
            } else {
                a.raw[4ll * q] = row.x; a.raw[4ll * q + 1] = row.y; a.raw[4ll * q + 2] = row.z;
            }
        }
    }
    report_range(a.status, amax_low, ADFP_STATUS_F16_RANGE_LOW, a.call_flag);
    report_range(a.status, amax_col, ADFP_STATUS_F16_RANGE_COLOR, a.call_flag);
}

// =============================================================================================
// The HIGH decoder on the in-band list (inference) in the G layout: DecodeArgs is k_decode_h's, a.packed points at the G image.
// A tile is 32 consecutive LIST entries; the front point of a lane is entry 16 (g >> 1) + n of its tile.
// =============================================================================================
template <int NT>
__global__ __launch_bounds__(NT, NT / 256) void k_decode_high_g(DecodeArgs a) {
#ifdef ADFP_EXP_HIGH_AS_LOW        // timing experiment (tools/inband_bisect.sh): the LOW network's body (32 grid channels, one gather) in THIS
    using L = DecLayoutG<32, 1>;   // kernel's launch shape -- list-indexed tiles, one image in LDS, 512 threads; a.packed = the low G image
#else
    using L = DecLayoutG<64, 1>;
#endif
    __shared__ __attribute__((aligned(16))) unsigned ldsu[L::P_TOTAL];
    __shared__ int s_next;
    __shared__ unsigned long long s_ring[ADFP_POOL_RING];
    image_to_lds<NT, L::P_TOTAL / 4>(ldsu, a.packed);
    if (threadIdx.x == 0) s_next = NT / 64;
    if (threadIdx.x < ADFP_POOL_RING) s_ring[threadIdx.x] = 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int count = *a.count_ptr;
    const int ntiles = (count + 31) >> 5;
    const TilePlan plan = tile_plan(ntiles, (int)gridDim.x, NT / 64, a.pool != nullptr);
    float amax = image_out_of_range<L::P_FLAG, L::NFLAG>(ldsu) ? INFINITY : 0.f;
#ifdef ADFP_STAMPS_G
    unsigned long long ph_[8] = {}, last_ = clock64();
#endif
    for (int j = threadIdx.x >> 6, tile; (tile = claim_tile_pool<NT / 64>(j, &s_next, s_ring, plan, ntiles, a.pool, a.status)) >= 0;) {
        const int idx = tile * 32 + 16 * (g >> 1) + n;
        const bool valid = idx < count;
#ifdef ADFP_HIGH_IDENTITY      // timing experiment: the same number of tiles over CONTIGUOUS points (no list indirection; results are not the network's)
        const int q = valid ? idx : 0;
#else
        const int q = a.list[valid ? idx : 0];
#endif
        float pn[3], pf[2][3];
        bool pnan;
        // the point's low-decoder value (added to the output at the end of the tile): requested NOW -- waited for after the network
        // it was a full memory latency per tile with nothing in flight (13 % of the wave's tile time, tools/phase_g.py)
        const float low_occ = (a.single || !valid) ? 0.f : a.raw[4ll * q + 3];
        {
            double pt[3];
            load_point(a.P, q, pt);
            normalize3(a.nb, pt, pn);
            const float f0 = (float)pt[0], f1 = (float)pt[1], f2 = (float)pt[2];
            pnan = (pt[0] != pt[0]) | (pt[1] != pt[1]) | (pt[2] != pt[2]);
            pf[0][0] = f0; pf[0][1] = f1; pf[0][2] = f2; pf[1][0] = f0; pf[1][1] = f1; pf[1][2] = f2;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) swap_halves(pf[0][k], pf[1][k]);
        float out[2][1];
#ifdef ADFP_STAMPS_G
        ph_[6] += 1;
#endif
        ADFP_PHG(0);
#ifdef ADFP_EXP_HIGH_AS_LOW
        decode_net_g<32, 1>(ldsu, a.g1, a.g1, pn, pf, lane, amax, out, nullptr, nullptr, 0u ADFP_PHG_ARGS(0));
#else
        decode_net_g<64, 1>(ldsu, a.g0, a.g1, pn, pf, lane, amax, out, nullptr, nullptr, 0u ADFP_PHG_ARGS(0));
#endif
        if (valid && (g & 1) == 0) {
            const float o = g >> 1 ? out[1][0] : out[0][0];
            const float v = pnan ? __builtin_nanf("") : o;
            a.att_occ[idx] = a.single ? v : v + low_occ;    // high + low, decoder.py:342
        }
        ADFP_PHG(5);
    }
#ifdef ADFP_STAMPS_G
#if ADFP_STAMPS_G != 2
    if (lane == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_phase_g[k], ph_[k]);
#endif
#endif
    report_range(a.status, amax, ADFP_STATUS_F16_RANGE_HIGH, a.call_flag);
}

// =============================================================================================
// attention fusion mlp_tsdf (a11), inference, in the G layout: 2 -> 64 -> 128 -> 128 -> 64 -> 2, softmax, convex blend.
// A layer's 32-unit out-blocks are the next layer's K = 32 groups (unit 32 b + unit16(8 g + j) in K-slot j of group b).
// =============================================================================================
struct AttLayoutG {
    using F = AttLayout;
    static constexpr int P_A0 = 0;                               // [64 rows in K order][4] f32 = (w0, w1, b, 0)
    static constexpr int P_W1 = 256;                             // 4 out-blocks x 2 K-groups
    static constexpr int P_B1 = P_W1 + 4 * 2 * 1024;             // [128] f32, unit order
    static constexpr int P_W2 = P_B1 + 128;                      // 4 out-blocks x 4 K-groups
    static constexpr int P_B2 = P_W2 + 4 * 4 * 1024;
    static constexpr int P_W3 = P_B2 + 128;                      // 2 out-blocks x 4 K-groups
    static constexpr int P_B3 = P_W3 + 2 * 4 * 1024;
    static constexpr int P_WO = P_B3 + 64;                       // [2 outputs][64] f32, unit order
    static constexpr int P_BO = P_WO + 128;
    static constexpr int P_FLAG = P_BO + 4;
    static constexpr int NFLAG = (((P_FLAG + 511) / 256) + 3) & ~3;
    static constexpr int P_TOTAL = P_FLAG + NFLAG;
    static_assert((P_TOTAL + 255) / 256 <= NFLAG, "one flag word per pack block");
};

__device__ HSrc att_g_src(int t) {
    using L = AttLayoutG;
    using F = AttLayout;
    if (t < L::P_W1) {
        const int row = t >> 2, c = t & 3;                       // row = 32 b + k: unit 32 b + unit16(k)
        const int u = (row & ~31) + unit16(row & 31);
        return HSrc{0, c < 2 ? F::F_W0 + u * 2 + c : (c == 2 ? F::F_B0 + u : -1), -1};
    }
    auto chain = [](int u, int ngrp, int base, int ld) {
        const int blk = u / (ngrp * 1024), v = u % (ngrp * 1024);
        const int grp = v >> 10, ob = (v >> 9) & 1, part = (v >> 8) & 1, g = (v >> 6) & 3, row = 32 * blk + 16 * ob + ((v >> 2) & 15), jp = (v & 3) * 2;
        return HSrc{1 + part, base + row * ld + 32 * grp + unit16(8 * g + jp), base + row * ld + 32 * grp + unit16(8 * g + jp + 1)};
    };
    if (t < L::P_B1) return chain(t - L::P_W1, 2, F::F_W1, 64);
    if (t < L::P_W2) return HSrc{0, F::F_B1 + (t - L::P_B1), -1};
    if (t < L::P_B2) return chain(t - L::P_W2, 4, F::F_W2, 128);
    if (t < L::P_W3) return HSrc{0, F::F_B2 + (t - L::P_B2), -1};
    if (t < L::P_B3) return chain(t - L::P_W3, 4, F::F_W3, 128);
    if (t < L::P_WO) return HSrc{0, F::F_B3 + (t - L::P_B3), -1};
    if (t < L::P_BO) return HSrc{0, F::F_WO + (t - L::P_WO), -1};
    const int o = t - L::P_BO;
    return HSrc{0, o < 2 ? F::F_BO + o : -1, -1};
}

ADFP_DEV void pack_attention_g_block(int blk, const float* __restrict__ flat, unsigned* __restrict__ packed, int* __restrict__ status) {
    using L = AttLayoutG;
    const int t = blk * 256 + (int)threadIdx.x;
    HSrc s{0, -1, -1};
    float a = 0.f, b = 0.f;
    if (t < L::P_FLAG) {
        s = att_g_src(t);
        a = s.s0 < 0 ? 0.f : flat[s.s0]; b = s.s1 < 0 ? 0.f : flat[s.s1];
    }
    pack_block_flag(!(fmaxf(fabsf(a), fabsf(b)) < 65504.0f), packed + L::P_FLAG + blk, status, ADFP_STATUS_F16_RANGE_ATT);
    if (t >= L::P_FLAG) return;
    if (s.kind == 0) { packed[t] = s.s0 < 0 ? 0u : __float_as_uint(a); return; }
    a = f16_clamp(a); b = f16_clamp(b);
    const float ah = f16_hi_part(a), bh = f16_hi_part(b);
    _Float16 x, y;
    if (s.kind == 1) { x = (_Float16)ah; y = (_Float16)bh; }
    else { x = (_Float16)(a - ah); y = (_Float16)(b - bh); }
    packed[t] = (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__global__ void k_pack_attention_g(const float* __restrict__ flat, unsigned* __restrict__ packed, int* __restrict__ status) { pack_attention_g_block((int)blockIdx.x, flat, packed, status); }

// AttArgs is k_attention_h's; a.packed points at the G image
template <int NT>
__global__ __launch_bounds__(NT) void k_attention_g(AttArgs a) {
    using A = AttLayoutG;
    __shared__ __attribute__((aligned(16))) unsigned ldsu[A::P_TOTAL];
    __shared__ int s_next;
    __shared__ unsigned long long s_ring[ADFP_POOL_RING];
    image_to_lds<NT, A::P_TOTAL / 4>(ldsu, a.packed);
    if (threadIdx.x == 0) s_next = NT / 64;
    if (threadIdx.x < ADFP_POOL_RING) s_ring[threadIdx.x] = 0ull;
    __syncthreads();
    const float* lds = (const float*)ldsu;
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const unsigned* wl = ldsu + 4 * lane;
    const float* b4 = lds + 4 * g;
    const float* arow = lds + 32 * g;
    const int count = a.count_ptr ? *a.count_ptr : a.n_rows;
    const int ntiles = (count + 31) >> 5;
    const TilePlan plan = tile_plan(ntiles, (int)gridDim.x, NT / 64, a.pool != nullptr);
    float amax = image_out_of_range<A::P_FLAG, A::NFLAG>(ldsu) ? INFINITY : 0.f;
    for (int j = threadIdx.x >> 6, tile; (tile = claim_tile_pool<NT / 64>(j, &s_next, s_ring, plan, ntiles, a.pool, a.status)) >= 0;) {
        int idx[2]; bool valid[2]; float occ[2], u[2];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            idx[pb] = tile * 32 + 16 * pb + n;
            valid[pb] = idx[pb] < count;
            const int ii = valid[pb] ? idx[pb] : 0;
            occ[pb] = a.att_occ[ii]; u[pb] = a.att_u[ii];
        }
        // layer 0 (2 -> 64) on the VALU: the lane's 8 units of each of the two K-groups, for both its rows
        f16x8 xh[4][2], xl[4][2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float t0[8], t1[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const f32x4 t = *(const f32x4*)(arow + A::P_A0 + (32 * b + k) * 4);            // row 32 b + 8 g + k
                t0[k] = relu_f(fmaf(u[0], t.y, fmaf(occ[0], t.x, t.z)));
                t1[k] = relu_f(fmaf(u[1], t.y, fmaf(occ[1], t.x, t.z)));
            }
            split8(t0, xh[b][0], xl[b][0], amax);
            split8(t1, xh[b][1], xl[b][1], amax);
        }
        __builtin_amdgcn_sched_barrier(0);
        // layer 1: 64 -> 128
        f16x8 yh[4][2], yl[4][2];
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            f32x4g acc[2][2];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) { const f32x4 t = *(const f32x4*)(b4 + A::P_B1 + 32 * blk + 16 * ob); acc[ob][0] = t; acc[ob][1] = t; }
            mfma_chain_g<2>(acc, wl + A::P_W1 + blk * 2 * 1024, xh, xl);
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const float t[8] = {relu_f(acc[0][pb][0]), relu_f(acc[0][pb][1]), relu_f(acc[0][pb][2]), relu_f(acc[0][pb][3]),
                                    relu_f(acc[1][pb][0]), relu_f(acc[1][pb][1]), relu_f(acc[1][pb][2]), relu_f(acc[1][pb][3])};
                split8(t, yh[blk][pb], yl[blk][pb], amax);
            }
        }
        // layer 2: 128 -> 128
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            f32x4g acc[2][2];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) { const f32x4 t = *(const f32x4*)(b4 + A::P_B2 + 32 * blk + 16 * ob); acc[ob][0] = t; acc[ob][1] = t; }
            mfma_chain_g<4>(acc, wl + A::P_W2 + blk * 4 * 1024, yh, yl);
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const float t[8] = {relu_f(acc[0][pb][0]), relu_f(acc[0][pb][1]), relu_f(acc[0][pb][2]), relu_f(acc[0][pb][3]),
                                    relu_f(acc[1][pb][0]), relu_f(acc[1][pb][1]), relu_f(acc[1][pb][2]), relu_f(acc[1][pb][3])};
                split8(t, xh[blk][pb], xl[blk][pb], amax);
            }
        }
        // layer 3: 128 -> 64, output 64 -> 2 on the VALU in f32
        float l0[2] = {0.f, 0.f}, l1[2] = {0.f, 0.f};
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            f32x4g acc[2][2];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) { const f32x4 t = *(const f32x4*)(b4 + A::P_B3 + 32 * blk + 16 * ob); acc[ob][0] = t; acc[ob][1] = t; }
            mfma_chain_g<4>(acc, wl + A::P_W3 + blk * 4 * 1024, xh, xl);
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                const f32x4 w0 = *(const f32x4*)(b4 + A::P_WO + 32 * blk + 16 * ob), w1 = *(const f32x4*)(b4 + A::P_WO + 64 + 32 * blk + 16 * ob);
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = relu_f(acc[ob][pb][r]);
                        l0[pb] = fmaf(v, w0[r], l0[pb]);
                        l1[pb] = fmaf(v, w1[r], l1[pb]);
                    }
            }
        }
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            l0[pb] += __shfl_xor(l0[pb], 16); l0[pb] += __shfl_xor(l0[pb], 32);
            l1[pb] += __shfl_xor(l1[pb], 16); l1[pb] += __shfl_xor(l1[pb], 32);
        }
        // softmax over 2, convex blend (decoder.py:255-258): lanes g = 0 finish block 0's row, lanes g = 1 block 1's
        if (g < 2 && valid[g]) {
            const float a0l = l0[g] + lds[A::P_BO], a1l = l1[g] + lds[A::P_BO + 1];
            const float m = fmaxf(a0l, a1l);
            const float e0 = expf(a0l - m), e1 = expf(a1l - m);
            const float den = e0 + e1;
            const float a0 = e0 / den, a1 = e1 / den;
            const float fused = a0 * occ[g] + a1 * u[g];
            const int ii = idx[g];
            const int q = a.list ? a.list[ii] : ii;              // list == NULL: mlp_tsdf.forward on explicit (occ, u) rows
            const bool inb = !a.flags || (a.flags[q] & ADFP_F_INBOUND) != 0;
            a.raw[4ll * q + 3] = (inb || !a.apply_bound) ? fused : 100.f;   // Renderer.py:64
            a.w[q] = a1;
        }
    }
    report_range(a.status, amax, ADFP_STATUS_F16_RANGE_ATT, a.call_flag);
}
