// adfp_decode_p.h -- software-pipelined form of the f16-split decoder kernel (adfp_decode_h.h).
//
// Measured on MI355X (tools/micro/mfma_fill.hip, gpurun_out/r02_micro_fill.log), v_mfma_f32_32x32x16_f16 stream of one
// wave with F independent v_fma_f32 placed in every MFMA gap, 1 / 2 / 3 waves per SIMD:
//     SIMD cycles per MFMA:   F = 0: 32.8 / 32.4 / 32.2    F = 5: 34.3 / 33.0 / 32.7    F = 8: 42.5 / 41.2 / 40.7
//                             F = 12: 59.0 / 52.3 / 51.1   F = 14: 77.5 / 60.7 / 58.0
//   -> about five VALU instructions per MFMA gap are free, every further one costs ~2.9 cycles of the SIMD; a
//      VALU-only stretch costs ~2.5 cycles per instruction (3 waves) and leaves the matrix pipe idle; v_pk_fma_f32 as a
//      filler costs 4 cycles each plus ~14 per gap (packed f32 is an anti-lever next to MFMAs on gfx950).
// k_decode_h runs a tile as  [~1000 VALU: point, gather, Fourier features]  then  [90 MFMA + ~300 VALU]: the first
// stretch leaves the matrix pipe to whatever the other waves of the SIMD happen to be doing.  Here a wave computes the
// features of its NEXT tile inside the MFMA gaps of the CURRENT tile: the loop body is one basic block holding
// features(t+1) and layers(t), two tiles' operand sets ping-pong in registers (2 waves per SIMD, 256 VGPRs).
#pragma once
#include "adfp_decode_h.h"

template <int CDIM>
struct TileFeat {                      // B operands of one tile: split grid features and Fourier features
    f16x8 ch[CDIM / 16], cl[CDIM / 16], eh[6], el[6];
};
struct TileMeta { int idx, q; bool valid, inb, pnan; };

template <int CDIM, int NOUT, int ROLE>
ADFP_DEV void p_features(const DecodeArgs& a, const float* __restrict__ lds, int tile, int count, int p, int h,
                         TileFeat<CDIM>& F, TileMeta& m, float& amax) {
    using L = DecLayoutH<CDIM, NOUT>;
    m.idx = tile * 32 + p;
    m.valid = m.idx < count;
    int q = m.valid ? m.idx : 0;
    if (ROLE == ROLE_HIGH) q = a.list[q];
    m.q = q;
    double pt[3]; float pn[3], pf[3];
    load_point(a.P, q, pt);
    normalize3(a.nb, pt, pn);
    pf[0] = (float)pt[0]; pf[1] = (float)pt[1]; pf[2] = (float)pt[2];   // p.float() decoder.py:189
    m.inb = in_bound(pt, a.b);
    m.pnan = (pt[0] != pt[0]) | (pt[1] != pt[1]) | (pt[2] != pt[2]);
    {
        float c[CDIM / 2];
        gather16(a.g0, pn, h, c);
        if (CDIM == 64) gather16(a.g1, pn, h, c + 16);
#pragma unroll
        for (int ks = 0; ks < L::KS_C; ++ks) split8(c + 8 * ks, F.ch[ks], F.cl[ks], amax);
    }
#pragma unroll
    for (int ks = 0; ks < L::KS_E; ++ks) {       // Fourier features sin(p @ B) (decoder.py:26-30)
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 bm = *(const f32x4*)(lds + L::P_BM + unit_of_h(ks, h, j) * 4);
            const float arg = fmaf(pf[2], bm.z, fmaf(pf[1], bm.y, pf[0] * bm.x));
            e[j] = adfp_sinf(arg);
        }
        split8<false>(e, F.eh[ks], F.el[ks], amax);     // |sin| <= 1
    }
}

// the same chain as mfma_chain_h without the scheduling fence behind it: the caller WANTS the surrounding VALU work
// of the next tile to move into the gaps
template <int NK>
ADFP_DEV void mfma_chain_p(f32x16& acc, const unsigned* __restrict__ w, int lane_off, const f16x8* __restrict__ xh, const f16x8* __restrict__ xl) {
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const f16x8 ah = __builtin_bit_cast(f16x8, *(const u32x4*)(w + ks * 512 + lane_off));
        const f16x8 al = __builtin_bit_cast(f16x8, *(const u32x4*)(w + ks * 512 + 256 + lane_off));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh[ks], acc, 0, 0, 0);
    }
}

// h = relu(W_i h + b_i) + (Wc_i c + bc_i); skip-concat [emb, h] feeds layer 3 (decoder.py:192-199)
template <int CDIM, int NOUT>
ADFP_DEV void p_layers(const unsigned* __restrict__ ldsu, int lane_off, int h, const TileFeat<CDIM>& F, f32x16& acc, float& amax) {
    using L = DecLayoutH<CDIM, NOUT>;
    const float* lds = (const float*)ldsu;
    f16x8 hh[2], hl[2];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        bias_init(acc, lds + L::P_BP(i), h);
        if (i == 0) mfma_chain_p<L::KS_E>(acc, ldsu + L::P_WP(0), lane_off, F.eh, F.el);
        else if (i == 3) {
            mfma_chain_p<L::KS_E>(acc, ldsu + L::P_WP(3), lane_off, F.eh, F.el);
            mfma_chain_p<2>(acc, ldsu + L::P_WP(3) + L::KS_E * 512, lane_off, hh, hl);
        } else mfma_chain_p<2>(acc, ldsu + L::P_WP(i), lane_off, hh, hl);
        relu_bias(acc, lds + L::P_BC(i), h);
        mfma_chain_p<L::KS_C>(acc, ldsu + L::P_WC(i), lane_off, F.ch, F.cl);
        if (i < 4) {
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = acc[r];
            split8(t, hh[0], hl[0], amax);
            split8(t + 8, hh[1], hl[1], amax);
        }
    }
}

template <int CDIM, int NOUT, int ROLE>
ADFP_DEV void p_output(const DecodeArgs& a, const float* __restrict__ lds, int h, const TileMeta& m, const f32x16& acc) {
    using L = DecLayoutH<CDIM, NOUT>;
    float out[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {             // output_linear on the VALU in f32: each half holds 16 of the 32 hidden units
        const float* wo = lds + L::P_WO + (h * NOUT + o) * 16;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s = fmaf(acc[r], wo[r], s);
        s += __shfl_xor(s, 32);
        out[o] = s + lds[L::P_BO + o];
        out[o] = m.pnan ? __builtin_nanf("") : out[o];          // see nan_point_outputs()
    }
    if (m.valid && h == 0) {
        const int q = m.q;
        if constexpr (ROLE == ROLE_LOW) {
            const unsigned f = a.flags ? a.flags[q] : 0u;
            a.raw[4ll * q + 3] = ((f & ADFP_F_BAND) || m.inb || !a.apply_bound) ? out[0] : 100.f;   // Renderer.py:64
            if (a.write_w) a.w[q] = 1.f;
        } else if constexpr (ROLE == ROLE_COLOR) {
            a.raw[4ll * q + 0] = out[0]; a.raw[4ll * q + 1] = out[1]; a.raw[4ll * q + 2] = out[2];
        } else {
            a.att_occ[m.idx] = out[0] + a.raw[4ll * q + 3];                        // high + low, decoder.py:342
        }
    }
}

template <int CDIM, int NOUT, int ROLE, int NT>
__global__ __launch_bounds__(NT, NT / 256) void k_decode_p(DecodeArgs a) {
    using L = DecLayoutH<CDIM, NOUT>;
    constexpr int NW = NT / 64;
    __shared__ __attribute__((aligned(16))) unsigned ldsu[L::P_TOTAL];
    __shared__ int s_next;
    for (int i = threadIdx.x; i < L::P_TOTAL / 4; i += NT) ((u32x4*)ldsu)[i] = ((const u32x4*)a.packed)[i];
    if (threadIdx.x == 0) s_next = NW;
    __syncthreads();
    const float* lds = (const float*)ldsu;
    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
    const int lane_off = h * 128 + p * 4;            // words: [h][32 rows][4 words = 8 halves]
    const int count = (ROLE == ROLE_HIGH) ? *a.count_ptr : a.P.n;
    const int ntiles = (count + 31) >> 5;
    float amax = 0.f;

    int j = threadIdx.x >> 6;
    int tile = claim_tile<NW>(j, &s_next, ntiles);
    if (tile >= 0) {
        TileFeat<CDIM> F0, F1;
        TileMeta m0, m1;
        f32x16 acc;
        p_features<CDIM, NOUT, ROLE>(a, lds, tile, count, p, h, F0, m0, amax);
        for (;;) {
            // ---- layers(F0) with features(next -> F1) in its gaps.  Without a next tile the features are recomputed for
            // the current one and dropped: one basic block either way (a branch would split the scheduling region).
            int next = claim_tile<NW>(j, &s_next, ntiles);
            __builtin_amdgcn_sched_barrier(0);
            p_features<CDIM, NOUT, ROLE>(a, lds, next >= 0 ? next : tile, count, p, h, F1, m1, amax);
            p_layers<CDIM, NOUT>(ldsu, lane_off, h, F0, acc, amax);
            __builtin_amdgcn_sched_barrier(0);
            p_output<CDIM, NOUT, ROLE>(a, lds, h, m0, acc);
            if (next < 0) break;
            tile = next;
            next = claim_tile<NW>(j, &s_next, ntiles);
            __builtin_amdgcn_sched_barrier(0);
            p_features<CDIM, NOUT, ROLE>(a, lds, next >= 0 ? next : tile, count, p, h, F0, m0, amax);
            p_layers<CDIM, NOUT>(ldsu, lane_off, h, F1, acc, amax);
            __builtin_amdgcn_sched_barrier(0);
            p_output<CDIM, NOUT, ROLE>(a, lds, h, m1, acc);
            if (next < 0) break;
            tile = next;
        }
    }
    report_range(a.status, amax);
}
