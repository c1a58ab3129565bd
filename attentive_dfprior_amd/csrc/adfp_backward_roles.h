// adfp_backward_roles.h -- the decoder backward WITH its weight gradients (adfp_backward_fused.h) re-cut so that it runs at TWO
// waves per SIMD instead of one.
//
// k_decode_bwd_fused keeps all 16 weight-gradient blocks of a 32-channel decoder (16 x 16 = 256 accumulation registers) in every
// wave, which pins the kernel to one wave per SIMD: 48 000 wave cycles per 32-point tile of which ~21 000 are issue -- every LDS
// round trip, every MFMA -> VALU hand-over and every DMA wait is exposed, and the colour decoder's backward ran at 0.038 of the
// f16 MFMA peak (round 4: 207 us of a 0.91 ms Mapper iteration).
//
// Here the 16 blocks are split by WHICH LAYER INPUT they are a product with, and a workgroup (512 threads, two waves per SIMD,
// 256 registers per lane) takes ONE of three roles for the whole launch:
//
//   role P ("Fourier")   dW_0 (3 blocks) and dW_3's Fourier part (3 blocks) against sin(p @ B), d embedder._B           7 blocks
//   role H ("hidden")    dW_1 .. dW_4 against h_0 .. h_3, output_linear (against h_4), the five pts_linears biases     5 blocks
//   role C ("features")  dWc_0 .. dWc_4 against the grid features c, the five fc_c biases, and d/d c (the grid gradient
//                        rows for k_scatter_sorted)                                                                      6 blocks
//
// Every role walks ALL tiles and runs the (cheap) cotangent chain gh_4 -> gh_0 itself -- 30 MFMAs and ~400 VALU instructions of
// the ~250 MFMAs and ~2 900 VALU instructions a tile cost the one-wave kernel -- but forms only its own products, so a wave needs
// at most 7 x 16 = 112 accumulation registers and two waves share a SIMD: the stalls of one are the other's issue slots.  The
// roles need no synchronisation with each other at all (different workgroups, different gradient elements); the expensive parts
// are not duplicated: the Fourier features are recomputed in role P only, every layer input is read (by LDS-DMA, once) by the
// one role that multiplies with it -- c by C, h_0 .. h_4 by H, nothing by P.  Workgroups are dealt to the roles in proportion
// to what a tile costs each (ROLE_SHARE_*).
//
// Everything else is the one-wave kernel's: the slot format of the transposition through LDS, the per-point power-of-two scale
// of the chain, the global scale S of the summed products, the narrow-product block with its column slots, the private copy of
// the flat gradient per workgroup (a role leaves the elements it does not own zero) reduced by k_reduce_partials_scaled.
// Same values up to the summation order over tiles (tests/test_gpu_grad.py::test_fused_weight_gradients_equal_the_staged_path).
#pragma once
#include "adfp_backward_fused.h"

// share of the workgroups per role, in 1/256 (what a tile costs a role: P ~ 8 100, H ~ 5 800, C ~ 6 200 SIMD cycles)
#ifndef ROLE_SHARE_P
#define ROLE_SHARE_P 104
#endif
#ifndef ROLE_SHARE_H
#define ROLE_SHARE_H 72
#endif

template <int NOUT, int ROLE>
__global__ __launch_bounds__(512) void k_decode_bwd_roles(DecodeBwdFArgs a, int nP, int nH) {
    constexpr int CDIM = 32;
    using LT = DecLayoutHT<CDIM, NOUT>;
    using ST = DecStage<CDIM>;
    using F = DecLayout<CDIM, NOUT>;
    static_assert(ROLE == ROLE_LOW || ROLE == ROLE_COLOR, "32-channel decoders only");
    // ONE shared array: the T image, then per wave 3 slots of 1 024 words (4 KB: one 32 x 32 block in the transposition format) and
    // 64 spare words.  Slot 0 is always the wave's transposition slot S; slots 1, 2 are: role C the double buffer of the grid
    // features' X block, role H the ring of the h_i X blocks, role P the parked d/d pre_3 block (S3) and the position table.
    constexpr int SLOT = 1024, NWV = 8, XW = 3136;
    static_assert(LT::P_TOTAL + NWV * XW <= 40960, "160 KB of LDS");
    static_assert(F::F_TOTAL <= NWV * XW, "the reduction copy must fit the per-wave regions");
    __shared__ __attribute__((aligned(16))) unsigned ldsu[LT::P_TOTAL + NWV * XW];
    const int bid = (int)blockIdx.x, nwg = (int)gridDim.x;
    const int role = bid < nP ? 0 : (bid < nP + nH ? 1 : 2);                 // block-uniform (scalar)
#ifdef ADFP_EXP_ONLY_ROLE          // timing experiment (tools/ab_roles.sh): only one role's workgroups do anything -- that role's own time at its share
    if (role != ADFP_EXP_ONLY_ROLE) return;
#endif
    // Role P never runs the fc_c^T chains: its copy of the image leaves the five T_WC blocks out (block (i, ib) of pts_linears^T
    // moves down by i + 1 blocks, PW below), and the 20 KB go to its waves: a third slot each (S0, layer 0's d/d pre).
    constexpr int P_CUT = 5 * 1024, XW_P = XW + P_CUT / NWV;
    if (role == 0) {
        for (int i = threadIdx.x; i < LT::P_TOTAL / 4; i += 512) {
            const int w = 4 * i;                                              // word offset in the full image
            int cut = 0; bool drop = false;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                if (w >= LT::T_WC(k) + 1024) cut += 1024;
                else if (w >= LT::T_WC(k)) drop = true;
            }
            if (!drop) *(u32x4*)(ldsu + w - cut) = ((const u32x4*)a.packed_t)[i];
        }
    } else {
        for (int i = threadIdx.x; i < LT::P_TOTAL / 4; i += 512) ((u32x4*)ldsu)[i] = ((const u32x4*)a.packed_t)[i];
    }
    __syncthreads();
    const float* lds = (const float*)ldsu;
    const int img_words = role == 0 ? LT::P_TOTAL - P_CUT : LT::P_TOTAL;
    const int xw = role == 0 ? XW_P : XW;
    float* s_red = (float*)(ldsu + img_words);

    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
    const int wvu = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane_off = h * 128 + p * 4;
    const int rwg = role == 0 ? bid : (role == 1 ? bid - nP : bid - nP - nH);
    const int nrwg = role == 0 ? nP : (role == 1 ? nH : nwg - nP - nH);
    const int wave = rwg * NWV + wvu, nwaves = nrwg * NWV;
    const int ntiles = (a.total + 31) >> 5;
    float amax = 0.f;
    const float gS = grad_scale(a.gmax);

    unsigned* xs = ldsu + img_words + wvu * xw;
    unsigned* slotS = xs;
    const unsigned xs_addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)xs;
    // ---- the slot format (adfp_backward_fused.h): piece (q4, hh) of a block = units 8 q4 + 4 hh .. + 3 of all 32 points, point pt in
    // lane slot hh * 32 + (pt ^ (2 q4 + hh)); the transposed read (lane = unit, registers = points kmapH(r, h)) is conflict-free
    int rbase[4];
    {
        const int q4j = p >> 3, hhj = (p >> 2) & 1, ej = p & 3, cj = 2 * q4j + hhj;
#pragma unroll
        for (int k = 0; k < 4; ++k) rbase[k] = q4j * 256 + (hhj * 32 + ((k + 4 * h) ^ cj)) * 4 + ej;
    }
    auto read_T = [&](const unsigned* slot, float* v) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = __uint_as_float(slot[rbase[r & 3] + 32 * (r >> 2)]);
    };
    auto write_blk = [&](unsigned* slot, const auto& v, float s) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
            *(f32x4*)(slot + q4 * 256 + (h * 32 + (p ^ (2 * q4 + h))) * 4) = f32x4{v[4 * q4] * s, v[4 * q4 + 1] * s, v[4 * q4 + 2] * s, v[4 * q4 + 3] * s};
    };
    // an X block (32 columns of the forward's layer-input rows from column `col`) of tile `tile_n` into slot `slot` by LDS-DMA, already
    // in the slot format (the XOR is applied to the SOURCE row a lane fetches).  4 VMEM operations, counted by hand below.
    auto dma_x = [&](int slot, int col, int tile_n) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int locn = tile_n * 32 + (p ^ (2 * q4 + h));
            const float* src = a.act + (long long)(locn < a.total ? locn : 0) * ST::NXM + col + 8 * q4 + 4 * h;
            unsigned keep;
            const unsigned dst = __builtin_amdgcn_readfirstlane(xs_addr + (unsigned)((slot * SLOT + q4 * 256) * 4));
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    };
    auto operand = [&](const unsigned* slot, f16x8* th, f16x8* tl) {           // a block out of a slot as operand halves (lane = unit, k = points)
        float v[16];
        read_T(slot, v);
        split16v(v, th, tl, amax);
    };
    auto operand_x = [&](const unsigned* slot, f16x8* th, f16x8* tl) {         // a layer-input block: range-checked by the forward
        float v[16];
        read_T(slot, v);
        split16v<false>(v, th, tl, amax);
    };

    // the small per-point inputs of a tile, fetched one tile ahead into the same registers once the layers have consumed them
    struct Small { double pt[3]; unsigned mw[3]; float go[4]; };
    auto fetch_small = [&](int tile_n, Small& sm, bool with_point) {
        const int locn = tile_n * 32 + p;
        const int qn = locn < a.total ? locn : 0;
        if (with_point) load_point(a.P, qn, sm.pt);
        const unsigned* mrow = a.masks + ((long long)qn * 2 + h) * 3;
        sm.mw[0] = mrow[0]; sm.mw[1] = mrow[1]; sm.mw[2] = mrow[2];
        if (ROLE == ROLE_LOW) { sm.go[0] = a.g_raw[4ll * qn + 3]; sm.go[1] = 0.f; sm.go[2] = 0.f; }
        else { sm.go[0] = a.g_raw[4ll * qn]; sm.go[1] = a.g_raw[4ll * qn + 1]; sm.go[2] = a.g_raw[4ll * qn + 2]; }
        sm.go[3] = 0.f;
    };
    // the head of a tile, common to the roles: d/d h_4 = Wo^T d out with the per-point power-of-two scale of the chain
    auto head_at = [&](int p_wo, const float* go, f32x16& gh, float& sc, float& isc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = 0.f;
#pragma unroll
            for (int o = 0; o < NOUT; ++o) s = fmaf(lds[p_wo + (h * NOUT + o) * 16 + r], go[o], s);
            gh[r] = s;
        }
        sc = 1.f; isc = 1.f;
        float m = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(gh[r]));
        m = fmaxf(m, __shfl_xor(m, 32));
        if (m > 0.f) {
            int se = 127 + 4 + 127 - (int)((__float_as_uint(m) >> 23) & 0xFFu);
            se = se < 1 ? 1 : (se > 253 ? 253 : se);
            sc = __uint_as_float((unsigned)se << 23);
            isc = __uint_as_float((unsigned)(254 - se) << 23);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) gh[r] *= sc;
    };
    auto head = [&](const Small& cur, bool valid, float* go, f32x16& gh, float& sc, float& isc) {
#pragma unroll
        for (int o = 0; o < 4; ++o) go[o] = valid ? cur.go[o] : 0.f;
        head_at(LT::P_WO, go, gh, sc, isc);
    };
    auto through_relu = [&](const f32x16& gh, unsigned m, float* gp) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int keep = ((int)(m << (16 + r))) >> 31;                       // -1 where unit r was active
            gp[r] = __uint_as_float(__float_as_uint(gh[r]) & (unsigned)keep);
        }
    };
    // D layout of a product: lane (n = p, h) register r = [row kmapH(r, h)][column p]; adds one block into the workgroup's copy
    auto add_rows = [&](const f32x16& acc, int base, int row_stride, int ncols) {
        if (p < ncols) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_red[base + kmapH(r, h) * row_stride + p] += acc[r];
        }
    };

    Small cur;
    if (role == 2) {
        // =====================================================================================================================
        // role C: fc_c[i].weight = d/d h_i (x) c, fc_c[i].bias, d/d c.  Blocks 0-4 = the five products, 5 = the bias columns.
        // slots 1 / 2: the grid features of this / the next tile (double buffer).
        // =====================================================================================================================
        constexpr int COL_C = ST::xm(ST::SC);
        f32x16 acc[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        if (wave < ntiles) { dma_x(1, COL_C, wave); fetch_small(wave, cur, false); }
        int it = 0;
        for (int tile = wave; tile < ntiles; tile += nwaves, ++it) {
            const int loc = tile * 32 + p;
            const bool valid = loc < a.total;
            const int q = valid ? loc : 0;
            const bool more = tile + nwaves < ntiles;
            const int tnext = tile + nwaves;
            const int par = __builtin_amdgcn_readfirstlane(it & 1);
            const unsigned mw0 = valid ? cur.mw[0] : 0u, mw1 = valid ? cur.mw[1] : 0u, mw2 = valid ? cur.mw[2] : 0u;
            const unsigned mk[5] = {mw0, mw0 >> 16, mw1, mw1 >> 16, mw2};
            float go[4]; f32x16 gh; float sc, isc;
            head(cur, valid, go, gh, sc, isc);
            const float ssc = isc * gS;
            // this tile's c was requested a tile ago.  With d/d c rows wanted, the previous tile's four row stores are the youngest
            // operations (or they passed the small loads: c is older than all of them either way); otherwise nothing countable is
            if (a.gc_out && it > 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            f16x8 cTh[2], cTl[2];
            operand_x(xs + (1 + par) * SLOT, cTh, cTl);
            if (more) dma_x(2 - par, COL_C, tnext);                          // the other buffer: read out a tile ago
            f32x16 gc;
#pragma unroll
            for (int r = 0; r < 16; ++r) gc[r] = 0.f;
#pragma unroll
            for (int i = 4; i >= 0; --i) {
                f16x8 xh[2], xl[2], tTh[2], tTl[2];
                write_blk(slotS, gh, ssc);
                split16(gh, xh, xl, amax);
                mfma_chain_h<2>(gc, ldsu + LT::T_WC(i), lane_off, xh, xl);
                operand(slotS, tTh, tTl);
                outer_job(acc[i], tTh, tTl, cTh, cTl);
                rowsum_job(acc[5], tTh, tTl, p, FSLOT_BFC(i));
                if (i > 0) {
                    float gp[16];
                    through_relu(gh, mk[i], gp);
                    split16v(gp, xh, xl, amax);
                    f32x16 gn;
#pragma unroll
                    for (int r = 0; r < 16; ++r) gn[r] = 0.f;
                    mfma_chain_h<2>(gn, ldsu + LT::T_WP(i, i == 3 ? 3 : 0), lane_off, xh, xl);
                    gh = gn;
                }
            }
            if (more) fetch_small(tnext, cur, false);
            if (a.gc_out && valid) stage_block_scaled(a.gc_out + 32ll * q, 0, h, gc, isc);     // 4 stores (a tile has a valid point)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int i = threadIdx.x; i < F::F_TOTAL; i += 512) s_red[i] = 0.f;
        __syncthreads();
        // round k: wave w adds its block (w + k) mod 8 -- different waves, different blocks: no two waves touch the same elements
        for (int k = 0; k < NWV; ++k) {
            const int j = (wvu + k) & 7;
            if (j == 0) add_rows(acc[0], F::F_FC(0), CDIM, 32);
            else if (j == 1) add_rows(acc[1], F::F_FC(1), CDIM, 32);
            else if (j == 2) add_rows(acc[2], F::F_FC(2), CDIM, 32);
            else if (j == 3) add_rows(acc[3], F::F_FC(3), CDIM, 32);
            else if (j == 4) add_rows(acc[4], F::F_FC(4), CDIM, 32);
            else if (j == 5) {
                if (p >= 5 && p < 10) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s_red[F::F_FC(p - 5) + 32 * CDIM + kmapH(r, h)] += acc[5][r];
                }
            }
            __syncthreads();
        }
    } else if (role == 1) {
        // =====================================================================================================================
        // role H: pts_linears[i].weight against h_{i-1} (i = 1..4; layer 3: the h_2 columns), the five pts_linears biases,
        // output_linear.  Blocks 0-3 = layers 1-4, 4 = narrow columns (FSLOT_BPL, FSLOT_WO, FSLOT_BO).
        // slots 1 / 2: a ring of two X blocks.  A tile starts with h_4 in A and h_3 in B (A = slot 1 on even tiles), and refills:
        // h_4 used -> A <- h_2;  layer 4 uses h_3 -> B <- h_1;  layer 3 uses h_2 -> A <- h_0;  layer 2 uses h_1 -> B <- next h_4;
        // layer 1 uses h_0 -> A <- next h_3: every block is requested two uses ahead, and exactly ONE request (4 operations) is
        // younger than the block a use waits for.
        // =====================================================================================================================
        f32x16 acc[5];
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        if (wave < ntiles) { dma_x(1, ST::xm(ST::SH(4)), wave); dma_x(2, ST::xm(ST::SH(3)), wave); fetch_small(wave, cur, false); }
        int it = 0;
        for (int tile = wave; tile < ntiles; tile += nwaves, ++it) {
            const int loc = tile * 32 + p;
            const bool valid = loc < a.total;
            const bool more = tile + nwaves < ntiles;
            const int tnext = tile + nwaves;
            const int par = __builtin_amdgcn_readfirstlane(it & 1);
            const int sA = 1 + par, sB = 2 - par;
            const unsigned mw0 = valid ? cur.mw[0] : 0u, mw1 = valid ? cur.mw[1] : 0u, mw2 = valid ? cur.mw[2] : 0u;
            const unsigned mk[5] = {mw0, mw0 >> 16, mw1, mw1 >> 16, mw2};
            float go[4]; f32x16 gh; float sc, isc;
            head(cur, valid, go, gh, sc, isc);
            const float ssc = isc * gS;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // h_4, h_3 (requested two uses ago) and the small inputs
            // ---------------- output_linear: d out (x) h_4 and its bias ----------------
            {
                float gob[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = 0.f;
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) v = (kmapH(r, h) == FSLOT_WO(o)) ? go[o] : v;
                    gob[r] = v;
                }
                write_blk(slotS, gob, gS);
                f16x8 gTh[2], gTl[2], hTh[2], hTl[2];
                operand(slotS, gTh, gTl);
                operand_x(xs + sA * SLOT, hTh, hTl);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the reads have returned before the DMA may overwrite the slot
                dma_x(sA, ST::xm(ST::SH(2)), tile);
                outer_job(acc[4], hTh, hTl, gTh, gTl);                        // [row = h_4 unit][column FSLOT_WO(o)]
                rowsum_job(acc[4], gTh, gTl, p, FSLOT_BO);                    // [row FSLOT_WO(o)][column FSLOT_BO]
            }
#pragma unroll
            for (int i = 4; i >= 0; --i) {
                f16x8 xh[2], xl[2], tTh[2], tTl[2];
                float gp[16];
                through_relu(gh, mk[i], gp);
                write_blk(slotS, gp, ssc);
                f32x16 gn;
                if (i > 0) {                                                  // the chain towards layer i - 1, in flight while the slot is read back
                    split16v(gp, xh, xl, amax);
#pragma unroll
                    for (int r = 0; r < 16; ++r) gn[r] = 0.f;
                    mfma_chain_h<2>(gn, ldsu + LT::T_WP(i, i == 3 ? 3 : 0), lane_off, xh, xl);
                }
                operand(slotS, tTh, tTl);
                rowsum_job(acc[4], tTh, tTl, p, FSLOT_BPL(i));
                if (i > 0) {
                    // layer 4: h_3 in B (waited for at the head of the tile); 3: h_2 in A; 2: h_1 in B; 1: h_0 in A
                    const int sl = (i & 1) ? sA : sB;
                    if (i < 4) {
                        if (i > 1 || more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    f16x8 hTh[2], hTl[2];
                    operand_x(xs + sl * SLOT, hTh, hTl);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (i == 4) dma_x(sl, ST::xm(ST::SH(1)), tile);
                    else if (i == 3) dma_x(sl, ST::xm(ST::SH(0)), tile);
                    else if (more) dma_x(sl, ST::xm(ST::SH(i == 2 ? 4 : 3)), tnext);
                    outer_job(acc[i - 1], tTh, tTl, hTh, hTl);
                    gh = gn;
                }
            }
            if (more) fetch_small(tnext, cur, false);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int i = threadIdx.x; i < F::F_TOTAL; i += 512) s_red[i] = 0.f;
        __syncthreads();
        for (int k = 0; k < NWV; ++k) {
            const int j = (wvu + k) & 7;
            if (j == 0) add_rows(acc[0], F::F_PL(1), 32, 32);
            else if (j == 1) add_rows(acc[1], F::F_PL(2), 32, 32);
            else if (j == 2) add_rows(acc[2], F::F_PL(3) + 93, 125, 32);
            else if (j == 3) add_rows(acc[3], F::F_PL(4), 32, 32);
            else if (j == 4) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int u = kmapH(r, h);
                    const float v = acc[4][r];
                    if (p < 5) s_red[F::F_PL(p) + 32 * F::in_dim(p) + u] += v;
                    else if (p >= 19 && p < 19 + NOUT) s_red[F::F_OW + (p - 19) * 32 + u] += v;
                    else if (p == FSLOT_BO && u >= 19 && u < 19 + NOUT) s_red[F::F_OB + (u - 19)] += v;
                }
            }
            __syncthreads();
        }
    } else {
        // =====================================================================================================================
        // role P: pts_linears[0].weight and the Fourier columns of pts_linears[3].weight against sin(p @ B), embedder._B through
        // cos(p @ B).  Blocks 0-2 = layer 0, 3-5 = layer 3, 6 = narrow columns FSLOT_EB.  No layer input is read: the chain runs
        // on masks and cotangents alone.  slot 1 = S3 (layer 3's d/d pre, parked until the Fourier blocks), then the position table.
        // =====================================================================================================================
        unsigned* slotS3 = xs + SLOT;
        unsigned* slotS0 = xs + 2 * SLOT;
        float* ptab = (float*)(xs + 3 * SLOT);
        static_assert(3 * SLOT + 128 <= XW_P, "role P: S, S3, S0 and the position table");
        auto PW = [](int i, int ib) { return LT::T_WP(i, ib) - 1024 * (i + 1); };      // block (i, ib) in the compacted image
        const int p_wo = LT::P_WO - P_CUT;
        f32x16 acc[7];
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        // a d/d pre block back out of its slot in the D layout, as the chain operand (stored x S / point scale, both powers of two)
        auto chain_operand = [&](const unsigned* slot, float back, f16x8* xh, f16x8* xl) {
            float t[16];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 v = *(const f32x4*)(slot + q4 * 256 + (h * 32 + (p ^ (2 * q4 + h))) * 4);
                t[4 * q4] = v.x * back; t[4 * q4 + 1] = v.y * back; t[4 * q4 + 2] = v.z * back; t[4 * q4 + 3] = v.w * back;
            }
            split16v<false>(t, xh, xl, amax);
        };
        if (wave < ntiles) fetch_small(wave, cur, true);
        for (int tile = wave; tile < ntiles; tile += nwaves) {
            const int loc = tile * 32 + p;
            const bool valid = loc < a.total;
            const bool more = tile + nwaves < ntiles;
            const int tnext = tile + nwaves;
            float pf[3] = {(float)cur.pt[0], (float)cur.pt[1], (float)cur.pt[2]};
            const bool pnan = (cur.pt[0] != cur.pt[0]) | (cur.pt[1] != cur.pt[1]) | (cur.pt[2] != cur.pt[2]);     // decoded at the origin by the forward
            if (pnan) { pf[0] = 0.f; pf[1] = 0.f; pf[2] = 0.f; }
            const unsigned mw0 = valid ? cur.mw[0] : 0u, mw1 = valid ? cur.mw[1] : 0u, mw2 = valid ? cur.mw[2] : 0u;
            const unsigned mk[5] = {mw0, mw0 >> 16, mw1, mw1 >> 16, mw2};
            float go[4]; f32x16 gh; float sc, isc;
#pragma unroll
            for (int o = 0; o < 4; ++o) go[o] = valid ? cur.go[o] : 0.f;
            head_at(p_wo, go, gh, sc, isc);
            const float ssc = isc * gS;
            if (h == 0) *(f32x4*)(ptab + 4 * p) = f32x4{pf[0], pf[1], pf[2], ssc};
#pragma unroll
            for (int i = 4; i >= 0; --i) {
                float gp[16];
                through_relu(gh, mk[i], gp);
                if (i == 3) write_blk(slotS3, gp, ssc);
                if (i == 0) write_blk(slotS0, gp, ssc);
                else {
                    f16x8 xh[2], xl[2];
                    split16v(gp, xh, xl, amax);
                    f32x16 gn;
#pragma unroll
                    for (int r = 0; r < 16; ++r) gn[r] = 0.f;
                    mfma_chain_h<2>(gn, ldsu + PW(i, i == 3 ? 3 : 0), lane_off, xh, xl);
                    gh = gn;
                }
            }
            if (more) fetch_small(tnext, cur, true);                          // in flight during the Fourier blocks
            // ---------------- the Fourier blocks, in the transposed layout: lane = feature 32 b + p, registers = the points kmapH(r, h) ----------------
            // the two d/d pre blocks as the products' left operands stay in registers across the three blocks; as CHAIN operands
            // (D layout) they are read back out of their slots per block -- 32 registers this role does not have
            f16x8 g0Th[2], g0Tl[2], g3Th[2], g3Tl[2];
            operand(slotS0, g0Th, g0Tl);
            operand(slotS3, g3Th, g3Tl);
            const float back = sc * (1.0f / gS);
            int pl = p;                                                       // opaque per tile: what depends on it is recomputed, not hoisted and spilled
            asm volatile("" : "+v"(pl));
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const f32x4 bm = *(const f32x4*)(lds + LT::P_BM + (32 * b + pl) * 4);
                const bool real = 32 * b + pl < 93;                           // the three padding features are not inputs
                float cs[16];
                f16x8 eTh[2], eTl[2];
                {
                    float e[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const f32x4 pv = *(const f32x4*)(ptab + 4 * kmapH(r, h));
                        const float arg = fmaf(pv.z, bm.z, fmaf(pv.y, bm.y, pv.x * bm.x));
                        float sn, c1;
                        adfp_sincosf(arg, sn, c1);
                        e[r] = real ? sn : 0.f;
                        cs[r] = c1 * pv.w;                                    // cos(p @ B) times the point's scale
                    }
                    split16v<false>(e, eTh, eTl, amax);
                }
                outer_job(acc[b], g0Th, g0Tl, eTh, eTl);
                outer_job(acc[3 + b], g3Th, g3Tl, eTh, eTl);
                // d/d (p @ B) = (W0_b^T gp_0 + W3_b^T gp_3) . cos(p @ B)
                f32x16 ge;
#pragma unroll
                for (int r = 0; r < 16; ++r) ge[r] = 0.f;
                {
                    f16x8 xh[2], xl[2];
                    chain_operand(slotS3, back, xh, xl);
                    mfma_chain_h<2>(ge, ldsu + PW(3, b), lane_off, xh, xl);
                    chain_operand(slotS0, back, xh, xl);
                    mfma_chain_h<2>(ge, ldsu + PW(0, b), lane_off, xh, xl);
                }
                write_blk(slotS, ge, 1.f);
                float ga[16];
                read_T(slotS, ga);
#pragma unroll
                for (int r = 0; r < 16; ++r) ga[r] *= cs[r];
                f16x8 aTh[2], aTl[2], bTh[2], bTl[2];
                split16v(ga, aTh, aTl, amax);
                {
                    // the positions as the B operand of the embedder._B products: lane FSLOT_EB(b, k) carries coordinate k of the 16 points
                    const int ks3 = pl - FSLOT_EB(b, 0);
                    const unsigned mx = ks3 == 0 ? ~0u : 0u, my = ks3 == 1 ? ~0u : 0u, mz = ks3 == 2 ? ~0u : 0u;
                    float pk[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const f32x4 pv = *(const f32x4*)(ptab + 4 * kmapH(r, h));
                        pk[r] = __uint_as_float((__float_as_uint(pv.x) & mx) | (__float_as_uint(pv.y) & my) | (__float_as_uint(pv.z) & mz));
                    }
                    split16v<false>(pk, bTh, bTl, amax);
                }
                outer_job(acc[6], aTh, aTl, bTh, bTl);                        // [row = feature 32 b + j][column FSLOT_EB(b, k)]
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int i = threadIdx.x; i < F::F_TOTAL; i += 512) s_red[i] = 0.f;
        __syncthreads();
        for (int k = 0; k < NWV; ++k) {
            const int j = (wvu + k) & 7;
            if (j < 3) {
                const int ncol = j < 2 ? 32 : 29;                             // 93 features
                if (j == 0) add_rows(acc[0], F::F_PL(0), 93, ncol);
                else if (j == 1) add_rows(acc[1], F::F_PL(0) + 32, 93, ncol);
                else add_rows(acc[2], F::F_PL(0) + 64, 93, ncol);
            } else if (j < 6) {
                const int ncol = j < 5 ? 32 : 29;
                if (j == 3) add_rows(acc[3], F::F_PL(3), 125, ncol);
                else if (j == 4) add_rows(acc[4], F::F_PL(3) + 32, 125, ncol);
                else add_rows(acc[5], F::F_PL(3) + 64, 125, ncol);
            } else if (j == 6) {
                if (p >= 10 && p < 19) {
                    const int bb = (p - 10) / 3, kk = (p - 10) % 3;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const int u = kmapH(r, h); if (32 * bb + u < 93) s_red[F::F_EB + kk * 93 + 32 * bb + u] += acc[6][r]; }
                }
            }
            __syncthreads();
        }
    }
    if (!(a.skip && *a.skip)) report_range(a.status, amax, ADFP_STATUS_F16_RANGE_BWD);
    float* part = a.partial + (long long)blockIdx.x * a.part_stride;
    for (int i = threadIdx.x; i < F::F_TOTAL; i += 512) part[i] = s_red[i];   // the slot is this workgroup's alone and written whole
}
