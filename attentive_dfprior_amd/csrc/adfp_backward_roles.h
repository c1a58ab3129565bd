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
// of the chain, the global scale S of the summed products, the narrow-product block with its column slots, the slot of partial
// sums per workgroup -- of which a workgroup here writes only its role's elements; k_reduce_partials_roles adds up each element over
// its owners' slots.
// Same values up to the summation order over tiles (tests/test_gpu_grad.py::test_fused_weight_gradients_equal_the_staged_path).
#pragma once
#include <type_traits>
#include "adfp_backward_fused.h"

// share of the workgroups per role, in 1/256: proportional to (time of the role alone) x (its workgroups) measured with the
// timing-only builds -DADFP_EXP_ONLY_ROLE=0/1/2 (tools/ab_roles.sh, profiles/r05_ab_backward_roles.txt)
#ifndef ROLE_SHARE_P
#define ROLE_SHARE_P 106
#endif
#ifndef ROLE_SHARE_H
#define ROLE_SHARE_H 78
#endif

#ifdef ADFP_STAMPS_ROLES           // debug build (tools/roles_span.py): per workgroup (role, wall-clock start, end of the tile loop, end), 100 MHz
__device__ unsigned long long g_roles_span[4 * 256];
#endif

template <int NOUT, int ROLE>
__global__ __launch_bounds__(512) void k_decode_bwd_roles(DecodeBwdFArgs a, int nP, int nH) {
#ifdef ADFP_STAMPS_ROLES
    const unsigned long long t_start_ = wall_clock64();
    unsigned long long t_loop_ = 0;
#endif
    constexpr int CDIM = 32;
    using LT = DecLayoutHT<CDIM, NOUT>;
    using ST = DecStage<CDIM>;
    using F = DecLayout<CDIM, NOUT>;
    static_assert(ROLE == ROLE_LOW || ROLE == ROLE_COLOR, "32-channel decoders only");
    // ONE shared array of 160 KB: the role's PART of the T image, then one region per wave.  A role copies only the blocks its
    // chains read, compacted, and its waves share what that frees:
    //   P  no fc_c^T blocks (10 of 15 blocks):  S, S3, S0 (4 KB slots: transposition scratch, layer 3's and layer 0's d/d pre) + position table
    //   H  the four main pts_linears^T blocks:   S + a ring of THREE X slots + the small-input buffer
    //   C  fc_c^T and the main pts_linears^T:    S + the double buffer of the grid features' X block + the small-input buffer
    constexpr int SLOT = 1024, NWV = 8, LDS_WORDS = 40960;
    constexpr int WO_WORDS = 2 * NOUT * 16;
    constexpr int IMG_P = LT::P_TOTAL - 5 * 1024, IMG_H = 4 * 1024 + WO_WORDS, IMG_C = 9 * 1024 + WO_WORDS;
    constexpr int XW_P = ((LDS_WORDS - IMG_P) / NWV) & ~3, XW_H = ((LDS_WORDS - IMG_H) / NWV) & ~3, XW_C = ((LDS_WORDS - IMG_C) / NWV) & ~3;
    constexpr int SMALL = 320;                                               // masks (32 x 6 words) + cotangent rows (32 x 4 floats) of one tile
    static_assert(3 * SLOT + 128 <= XW_P && 4 * SLOT + SMALL <= XW_H && 3 * SLOT + SMALL <= XW_C, "per-wave regions");
    static_assert(NWV * 3 * 16 * 64 <= LDS_WORDS, "the reduction's three blocks per wave");
    __shared__ __attribute__((aligned(16))) unsigned ldsu[LDS_WORDS];
    const int bid = (int)blockIdx.x, nwg = (int)gridDim.x;
#ifdef ADFP_EXP_COMPILE_ROLE       // ISA experiments: the kernel with one role's code only (tools/isa_mix.py per role)
    const int role = ADFP_EXP_COMPILE_ROLE;
#else
    const int role = bid < nP ? 0 : (bid < nP + nH ? 1 : 2);                 // block-uniform (scalar)
#endif
#ifdef ADFP_EXP_ONLY_ROLE          // timing experiment (tools/ab_roles.sh): only one role's workgroups do anything -- that role's own time at its share
    if (role != ADFP_EXP_ONLY_ROLE) return;
#endif
    auto copy_words = [&](int dst, int src, int n) {                         // n words of the packed image -> LDS (multiples of 4)
        image_to_lds<512>(ldsu + dst, a.packed_t + src, n / 4);
    };
    if (role == 0) {                                                         // block (i, ib) of pts_linears^T moves down by i + 1 blocks (PW below)
        copy_words(0, 0, LT::T_WC(0));
#pragma unroll
        for (int i = 0; i < 5; ++i) copy_words(LT::T_WP(i, 0) - 1024 * (i + 1), LT::T_WP(i, 0), LT::nb(i) * 1024);
        copy_words(LT::P_WO - 5 * 1024, LT::P_WO, WO_WORDS);
    } else if (role == 1) {                                                  // HW(i) = (i - 1) * 1024: the main block of layer i = 1..4
#pragma unroll
        for (int i = 1; i < 5; ++i) copy_words((i - 1) * 1024, LT::T_WP(i, i == 3 ? 3 : 0), 1024);
        copy_words(4 * 1024, LT::P_WO, WO_WORDS);
    } else {                                                                 // CWC(i) = i ? (2 i - 1) * 1024 : 0, CWP(i) = 2 i * 1024
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            copy_words(i ? (2 * i - 1) * 1024 : 0, LT::T_WC(i), 1024);
            if (i) copy_words(2 * i * 1024, LT::T_WP(i, i == 3 ? 3 : 0), 1024);
        }
        copy_words(9 * 1024, LT::P_WO, WO_WORDS);
    }
    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
    const int wvu = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int img_words = role == 0 ? IMG_P : (role == 1 ? IMG_H : IMG_C);
    const int xw = role == 0 ? XW_P : (role == 1 ? XW_H : XW_C);
    // Tiles of a workgroup are handed to its waves through an LDS ticket (the last word of wave 0's region, spare in every role): the
    // SIMD arbiter favours the older of its two waves, and with a fixed split the four old waves of a workgroup were done a fifth
    // of the launch before the four young ones, which then ran alone on their SIMDs (per-workgroup stamps, tools/roles_span.py).
    int* s_ticket = (int*)(ldsu + img_words + xw - 1);
    if (threadIdx.x == 0) *s_ticket = NWV;                                   // local tile numbers 0 .. 7 are the waves' first tiles
    __syncthreads();
    const float* lds = (const float*)ldsu;

    const int lane_off = h * 128 + p * 4;
    const int rwg = role == 0 ? bid : (role == 1 ? bid - nP : bid - nP - nH);
    const int nrwg = role == 0 ? nP : (role == 1 ? nH : nwg - nP - nH);
    const int ntiles = (a.total + 31) >> 5;
    // the workgroup's tiles: rwg, rwg + nrwg, ...; local number j <-> tile rwg + j nrwg.  A wave draws its NEXT tile at the top of a
    // tile (the answer takes an LDS round trip and is needed after the head, for the prefetches): it never sits on more than one
    // undone tile.  (Drawn at the END of a tile instead, the loop-carried draw made the register allocator spill 350 registers.)
    const int ntl = rwg < ntiles ? (ntiles - rwg + nrwg - 1) / nrwg : 0;
    auto claim = [&]() {
        int v = 0;
        if (lane == 0) v = atomicAdd(s_ticket, 1);
        return __builtin_amdgcn_readfirstlane(v);
    };
    auto tile_of = [&](int j) { return rwg + j * nrwg; };
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
    // in the slot format (the XOR is applied to the SOURCE row a lane fetches).  4 VMEM operations, counted by hand below.  Address =
    // a SCALAR base (the tile's first row + the column, 64-bit arithmetic on the scalar unit) + a 32-bit lane offset that does not
    // depend on the tile (four loop-invariant registers): per-lane 64-bit pointers cost ~8 VALU instructions per operation and the
    // registers that held their loop-invariant parts were spilled and reloaded every tile.  Rows beyond the end (last tile only) are
    // fetched from the tile's first row (finite; their cotangents are zero).
    int rowoff[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) rowoff[q4] = ((p ^ (2 * q4 + h)) * ST::NXM + 8 * q4 + 4 * h) * 4;
    auto lds_dma = [&](unsigned dst_bytes, int voff, const void* sbase, auto width) {     // width: 4 = dwordx4, 1 = dword
        unsigned keep;
        const unsigned dst = __builtin_amdgcn_readfirstlane(dst_bytes);
        if constexpr (decltype(width)::value == 4)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(dst), "s"(sbase) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(dst), "s"(sbase) : "memory");
    };
    auto dma_x = [&](int slot, int col, int tile_n) {
        const float* sbase = a.act + ((long long)tile_n * 32 * ST::NXM + col);
        const int rows_left = a.total - tile_n * 32;                            // scalar; >= 1
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            int off = rowoff[q4];
            if (rows_left < 32 && (p ^ (2 * q4 + h)) >= rows_left) off = (8 * q4 + 4 * h) * 4;
            lds_dma(xs_addr + (unsigned)((slot * SLOT + q4 * 256) * 4), off, sbase, std::integral_constant<int, 4>{});
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

    // Roles H and C: the small per-point inputs of a tile (ReLU mask words, cotangent of the decoder output) come through LDS as
    // well -- the tile's 32 mask rows (768 B) and 32 cotangent rows (512 B) are contiguous in memory: five 256-byte LDS-DMA
    // operations copy them verbatim into the wave's small-input buffer, requested right after the previous tile's head has read
    // the buffer out.  (As ordinary loads the compiler's own s_waitcnt for them drains every younger DMA request at each tile's
    // head -- it does not know those exist -- which stalled a wave for a memory latency per tile.)
    // (address arithmetic that is used once per tile starts from an OPAQUE copy of the lane index: hoisted out of the tile loop these
    // values are loop-invariant registers the kernel does not have -- they were spilled before the loop and re-loaded every tile)
    auto opaque_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
    auto dma_small = [&](unsigned* buf, int tile_n) {
        const int lane = opaque_lane();
        const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)buf;
        const unsigned* smask = a.masks + (long long)tile_n * 192;
        const float* sgo = a.g_raw + (long long)tile_n * 128;
        const int rows_left = a.total - tile_n * 32;                            // scalar; the last word a partial tile may read
        const int lim_m = (rows_left < 32 ? rows_left * 6 - 1 : 191) * 4, lim_g = (rows_left < 32 ? rows_left * 4 - 1 : 127) * 4;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            int off = (64 * (k < 3 ? k : k - 3) + lane) * 4;
            const int lim = k < 3 ? lim_m : lim_g;
            off = off < lim ? off : lim;
            lds_dma(base + (unsigned)(k * 256), off, k < 3 ? (const void*)smask : (const void*)sgo, std::integral_constant<int, 1>{});
        }
    };
    // the small per-point inputs of a tile, fetched one tile ahead into the same registers once the layers have consumed them
    struct Small { unsigned mw[3]; float go[4]; };
    auto read_small = [&](const unsigned* buf, Small& sm) {
        const int lane = opaque_lane(), p = lane & 31, h = lane >> 5;
        const unsigned* m = buf + (p * 2 + h) * 3;
        sm.mw[0] = m[0]; sm.mw[1] = m[1]; sm.mw[2] = m[2];
        const f32x4 g = *(const f32x4*)(buf + 192 + 4 * p);
        if (ROLE == ROLE_LOW) { sm.go[0] = g.w; sm.go[1] = 0.f; sm.go[2] = 0.f; }
        else { sm.go[0] = g.x; sm.go[1] = g.y; sm.go[2] = g.z; }
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
    // ---- the end of the launch: the eight waves' accumulators are added up and leave as this workgroup's slot of partial sums.
    // D layout of a product: lane (n = p, h) register r = [row kmapH(r, h)][column p].  Three blocks at a time go through LDS
    // (the whole array: nothing of the tile loop is live any more) as [wave][block][register][lane] -- 48 conflict-free ds_write_b32
    // per lane -- and after a barrier wave w adds up six of the 48 (block, register) rows over the eight waves, in wave order, and
    // stores them straight to memory: the two 128-byte runs of a row.  A workgroup writes ONLY the elements its role owns
    // (role_of_element below: k_reduce_partials_roles reads each element from its owners' slots only) -- a third of the bytes of a
    // whole-slot copy, written and read back.  (Eight rounds of read-modify-write on a whole-slot LDS copy + the copy-out were
    // ~12 us of every workgroup's launch.)
    float* const red = (float*)ldsu;
    float* const part = a.partial + (long long)blockIdx.x * a.part_stride;
    auto put = [&](int b, const f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wvu * 3 + b) * 16 + r) * 64 + lane] = acc[r];
    };
    auto rows = [&](auto&& dest) {
#pragma unroll 1
        for (int k = 0; k < 6; ++k) {
            const int row = wvu * 6 + k, b = row >> 4, r = row & 15;      // wave-uniform
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) v += red[((w * 3 + b) * 16 + r) * 64 + lane];
            dest(b, r, v);
        }
    };

#ifdef ADFP_EXP_ROLES_PRIO          // timing experiment: the YOUNG wave of every SIMD (waves 4-7 start later) at a raised issue priority
    if (wvu >= 4) __builtin_amdgcn_s_setprio(2);
#endif
    Small cur;
    if (role == 2) {
        // =====================================================================================================================
        // role C: fc_c[i].weight = d/d h_i (x) c, fc_c[i].bias, d/d c.  Blocks 0-4 = the five products, 5 = the bias columns.
        // slots 1 / 2: the grid features of this / the next tile (double buffer).
        // =====================================================================================================================
        constexpr int COL_C = ST::xm(ST::SC);
        auto CWC = [](int i) { return i ? (2 * i - 1) * 1024 : 0; };
        auto CWP = [](int i) { return 2 * i * 1024; };
        unsigned* small = xs + 3 * SLOT;
        f32x16 acc[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        int jc = wvu;
        if (jc < ntl) { dma_x(1, COL_C, tile_of(jc)); dma_small(small, tile_of(jc)); }
        int it = 0;
        for (; jc < ntl; ++it) {
            const int tile = tile_of(jc);
            const int jn = claim();                                          // the wave's NEXT tile: drawn first thing, needed after the head
            const bool more = jn < ntl;
            const int tnext = tile_of(jn);
            const int loc = tile * 32 + p;
            const bool valid = loc < a.total;
            const int q = valid ? loc : 0;
            const int par = __builtin_amdgcn_readfirstlane(it & 1);
            // this tile's c and small inputs were requested a tile ago; with d/d c rows wanted, the previous tile's four row stores
            // are the only younger operations
            if (a.gc_out && it > 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            read_small(small, cur);
            f16x8 cTh[2], cTl[2];
            operand_x(xs + (1 + par) * SLOT, cTh, cTl);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the small buffer has been read out
            if (more) { dma_x(2 - par, COL_C, tnext); dma_small(small, tnext); }   // the other c buffer: read out a tile ago
            const unsigned mw0 = valid ? cur.mw[0] : 0u, mw1 = valid ? cur.mw[1] : 0u, mw2 = valid ? cur.mw[2] : 0u;
            const unsigned mk[5] = {mw0, mw0 >> 16, mw1, mw1 >> 16, mw2};
            float go[4]; f32x16 gh; float sc, isc;
#pragma unroll
            for (int o = 0; o < 4; ++o) go[o] = valid ? cur.go[o] : 0.f;
            head_at(9 * 1024, go, gh, sc, isc);
            const float ssc = isc * gS;
            f32x16 gc;
#pragma unroll
            for (int r = 0; r < 16; ++r) gc[r] = 0.f;
#pragma unroll
            for (int i = 4; i >= 0; --i) {
                f16x8 xh[2], xl[2], tTh[2], tTl[2];
                write_blk(slotS, gh, ssc);
                split16(gh, xh, xl, amax);
                mfma_chain_h<2>(gc, ldsu + CWC(i), lane_off, xh, xl);
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
                    mfma_chain_h<2>(gn, ldsu + CWP(i), lane_off, xh, xl);
                    gh = gn;
                }
            }
            if (a.gc_out && valid) stage_block_scaled(a.gc_out + 32ll * q, 0, h, gc, isc);     // 4 stores (a tile has a valid point)
            jc = jn;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef ADFP_STAMPS_ROLES
        t_loop_ = wall_clock64();
#endif
        __syncthreads();
        put(0, acc[0]); put(1, acc[1]); put(2, acc[2]);
        __syncthreads();
        rows([&](int b, int r, float v) { part[F::F_FC(b) + kmapH(r, h) * CDIM + p] = v; });
        __syncthreads();
        put(0, acc[3]); put(1, acc[4]); put(2, acc[5]);
        __syncthreads();
        rows([&](int b, int r, float v) {
            if (b < 2) part[F::F_FC(3 + b) + kmapH(r, h) * CDIM + p] = v;
            else if (p >= 5 && p < 10) part[F::F_FC(p - 5) + 32 * CDIM + kmapH(r, h)] = v;       // the bias columns
        });
    } else if (role == 1) {
        // =====================================================================================================================
        // role H: pts_linears[i].weight against h_{i-1} (i = 1..4; layer 3: the h_2 columns), the five pts_linears biases,
        // output_linear.  Blocks 0-3 = layers 1-4, 4 = narrow columns (FSLOT_BPL, FSLOT_WO, FSLOT_BO).
        // =====================================================================================================================
        // The X ring: the five blocks of a tile are USED in the order h_4 (output_linear), h_3 (layer 4), h_2, h_1, h_0; use number
        // n = 5 it + k of the wave sits in slot 1 + n mod 3 and is requested when use n - 3 has read that slot out -- three uses =
        // about two layers of work ahead.  When a use waits, the requests of the next two uses (8 operations) are younger -- and, for
        // uses 0-2, the five operations of the next tile's small inputs, requested right after this tile's head (counting only 8
        // there made the wave wait for most of the NEXT block as well: a lead of two uses instead of three).  The small inputs are
        // older than all five of the tile's X requests (20 operations) when the next head waits for them.
        unsigned* small = xs + 4 * SLOT;
        f32x16 acc[5];
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        auto col_of = [](int k) { return ST::xm(ST::SH(4 - k)); };            // use k reads h_{4-k}
        int jc = wvu;
        if (jc < ntl) {
            const int t0 = tile_of(jc);
            dma_small(small, t0); dma_x(1, col_of(0), t0); dma_x(2, col_of(1), t0); dma_x(3, col_of(2), t0);
        }
        int it = 0;
        for (; jc < ntl; ++it) {
            const int tile = tile_of(jc);
            const int jn = claim();
            const bool more = jn < ntl;
            const int tnext = tile_of(jn);
            const int loc = tile * 32 + p;
            const bool valid = loc < a.total;
            const int n0 = __builtin_amdgcn_readfirstlane((5 * it) % 3);       // ring position of the tile's first use
            auto slot_of = [&](int k) { const int t = n0 + k; return 1 + (t >= 6 ? t - 6 : (t >= 3 ? t - 3 : t)); };
            // use k: wait for its block, read it out, then request use k + 3 (of this or the next tile) into the same slot
            auto take = [&](int k, f16x8* hTh, f16x8* hTl) {
                const bool two = more || k < 3, one = more || k < 4;            // are the requests of the next two / one uses out?
                // uses 0-2: the next tile's small inputs (5 operations, requested after this tile's head) are younger as well
#ifndef ADFP_EXP_H_NOWAIT          // timing experiment: the blocks are not waited for (what the memory waits of this role cost)
                if (k < 3 && more) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
                else if (two) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (one) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                const int sl = slot_of(k);
                operand_x(xs + sl * SLOT, hTh, hTl);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the reads have returned before the DMA may overwrite the slot
                if (k < 2) dma_x(sl, col_of(k + 3), tile);
                else if (more) dma_x(sl, col_of(k - 2), tnext);
            };
            // the small inputs: requested a tile ago, 20 X-request operations younger (first tile: 12)
            if (it > 0) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            read_small(small, cur);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (more) dma_small(small, tnext);
            const unsigned mw0 = valid ? cur.mw[0] : 0u, mw1 = valid ? cur.mw[1] : 0u, mw2 = valid ? cur.mw[2] : 0u;
            const unsigned mk[5] = {mw0, mw0 >> 16, mw1, mw1 >> 16, mw2};
            float go[4]; f32x16 gh; float sc, isc;
#pragma unroll
            for (int o = 0; o < 4; ++o) go[o] = valid ? cur.go[o] : 0.f;
            head_at(4 * 1024, go, gh, sc, isc);
            const float ssc = isc * gS;
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
                take(0, hTh, hTl);
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
                    mfma_chain_h<2>(gn, ldsu + (i - 1) * 1024, lane_off, xh, xl);
                }
                operand(slotS, tTh, tTl);
                rowsum_job(acc[4], tTh, tTl, p, FSLOT_BPL(i));
                if (i > 0) {
                    f16x8 hTh[2], hTl[2];
                    take(5 - i, hTh, hTl);                                    // layer i multiplies with h_{i-1}: use 5 - i
                    outer_job(acc[i - 1], tTh, tTl, hTh, hTl);
                    gh = gn;
                }
            }
            jc = jn;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef ADFP_STAMPS_ROLES
        t_loop_ = wall_clock64();
#endif
        __syncthreads();
        put(0, acc[0]); put(1, acc[1]); put(2, acc[2]);
        __syncthreads();
        rows([&](int b, int r, float v) {
            const int base = b == 0 ? F::F_PL(1) : (b == 1 ? F::F_PL(2) : F::F_PL(3) + 93), stride = b == 2 ? 125 : 32;
            part[base + kmapH(r, h) * stride + p] = v;
        });
        __syncthreads();
        put(0, acc[3]); put(1, acc[4]);
        __syncthreads();
        rows([&](int b, int r, float v) {
            if (b == 0) part[F::F_PL(4) + kmapH(r, h) * 32 + p] = v;
            else if (b == 1) {
                // column p of the narrow block is a slot: one destination row per lane (or none), element kmapH(r, h) of it
                const int dst = p < 5 ? F::F_PL(p) + 32 * F::in_dim(p) : ((p >= 19 && p < 19 + NOUT) ? F::F_OW + (p - 19) * 32 : -1);
                const int u = kmapH(r, h);
                if (dst >= 0) part[dst + u] = v;
                else if (p == FSLOT_BO && u >= 19 && u < 19 + NOUT) part[F::F_OB + (u - 19)] = v;
            }                                                                 // (the third block of this round holds nothing)
        });
    } else {
        // =====================================================================================================================
        // role P: pts_linears[0].weight and the Fourier columns of pts_linears[3].weight against sin(p @ B) -- blocks 0-2 = layer 0,
        // 3-5 = layer 3 -- and embedder._B through cos(p @ B).  No layer input is read: the chain runs on masks and cotangents alone.
        // slots: S (transposition scratch), S3 and S0 (layer 3's / layer 0's d/d pre: written by the chain, read in the Fourier
        // blocks), then the position table (32 x {x, y, z, scale}) and the small-input buffer (masks, cotangents, z, the tile's rays).
        //
        // d embedder._B[k][j] = sum_p x_k(p) cos(p @ B)_j ge_j(p) is formed on the VALU: in the transposed layout a lane IS feature j
        // and holds ge_j of 16 points, so the sum over the tile's points is 16 fma per coordinate into 9 f32 accumulators per lane
        // (3 feature blocks x 3 coordinates) that live across all tiles.  (The one-wave kernel sent it through the matrix pipe: a
        // split of the 16 values, a masked positions operand per block and 6 MFMAs into a narrow accumulator block -- 32 + 72
        // quarter-rate instructions and 16 registers more than 48 full-rate fma; this role is issue bound on exactly those.)
        // =====================================================================================================================
        unsigned* slotS3 = xs + SLOT;
        unsigned* slotS0 = xs + 2 * SLOT;
        float* ptab = (float*)(xs + 3 * SLOT);
        unsigned* small = xs + 3 * SLOT + 128;
        static_assert(3 * SLOT + 128 + SMALL + 192 <= XW_P, "role P: three slots, the position table, the small-input buffer with the positions' sources");
        auto PW = [](int i, int ib) { return LT::T_WP(i, ib) - 1024 * (i + 1); };      // block (i, ib) in the compacted image
        const int p_wo = LT::P_WO - 5 * 1024;
        f32x16 acc[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        float eb[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};          // [feature block][coordinate]: this lane's feature, its half's points
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
        auto park = [&](unsigned* slot, const f16x8* xh, const f16x8* xl) {
            const int lane = opaque_lane();
            *(u32x4*)(slot + (0 * 64 + lane) * 4) = __builtin_bit_cast(u32x4, xh[0]);
            *(u32x4*)(slot + (1 * 64 + lane) * 4) = __builtin_bit_cast(u32x4, xh[1]);
            *(u32x4*)(slot + (2 * 64 + lane) * 4) = __builtin_bit_cast(u32x4, xl[0]);
            *(u32x4*)(slot + (3 * 64 + lane) * 4) = __builtin_bit_cast(u32x4, xl[1]);
        };
        auto unpark = [&](const unsigned* slot, f16x8* xh, f16x8* xl) {
            xh[0] = __builtin_bit_cast(f16x8, *(const u32x4*)(slot + (0 * 64 + lane) * 4));
            xh[1] = __builtin_bit_cast(f16x8, *(const u32x4*)(slot + (1 * 64 + lane) * 4));
            xl[0] = __builtin_bit_cast(f16x8, *(const u32x4*)(slot + (2 * 64 + lane) * 4));
            xl[1] = __builtin_bit_cast(f16x8, *(const u32x4*)(slot + (3 * 64 + lane) * 4));
        };
        // The tile's positions come through LDS as well (three more DMA operations, 192 words): in ray mode z_vals of its 32 points
        // (64 words) and the up to five rays those points can belong to (S >= 8; lanes 0-29 fetch origin + direction of rays
        // r0 .. r0 + 4); explicit points (adfp_eval_points_backward) are the tile's 32 rows verbatim, 192 (f64) or 96 (f32) words.
        auto dma_pos = [&](unsigned* buf, int tile_n) {
            const int lane = opaque_lane();
            const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)buf;
            const int rows_left = a.total - tile_n * 32;
            if (a.P.mode == ADFP_PTS_RAYS) {
                // z: 32 doubles = 64 words verbatim; rays: lane = 6 ray + component of the up to five rays from the tile's first on
                const unsigned* sz = (const unsigned*)a.P.z + (long long)tile_n * 64;
                const int lim = (rows_left < 32 ? rows_left * 2 - 1 : 63) * 4;
                int off = lane * 4;
                off = off < lim ? off : lim;
                lds_dma(base, off, sz, std::integral_constant<int, 1>{});
                const int r0 = (int)((unsigned)(tile_n * 32) / (unsigned)a.P.S);
                const int nrays = a.P.n / a.P.S;
                const int rr = lane < 30 ? lane / 6 : 0, cc = lane < 30 ? lane % 6 : 0;
                const int ray = (r0 + rr < nrays ? r0 + rr : nrays - 1) - r0;       // relative to the scalar base
                const int roff = (3 * ray + (cc < 3 ? cc : cc - 3)) * 4;
                const float* so = a.P.ro + 3ll * r0;
                const float* sd = a.P.rd + 3ll * r0;
                // origin and direction live in two arrays: two operations, each lane takes its word from the right one
                lds_dma(base + 256u, cc < 3 ? roff : 0, so, std::integral_constant<int, 1>{});
                lds_dma(base + 512u, cc < 3 ? 0 : roff, sd, std::integral_constant<int, 1>{});
            } else {
                const int wpp = a.P.mode == ADFP_PTS_F64 ? 6 : 3;                // words per point
                const unsigned* sp = (const unsigned*)a.P.pts + (long long)tile_n * 32 * wpp;
                const int lim = ((rows_left < 32 ? rows_left : 32) * wpp - 1) * 4;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    int off = (64 * k + lane) * 4;
                    off = off < lim ? off : lim;
                    lds_dma(base + (unsigned)(k * 256), off, sp, std::integral_constant<int, 1>{});
                }
            }
        };
        auto read_pos = [&](const unsigned* buf, int tile_n, double* pt) {
            const int p = opaque_lane() & 31;
            if (a.P.mode == ADFP_PTS_RAYS) {
                const unsigned q0 = (unsigned)tile_n * 32u;
                const int rr = (int)((q0 + (unsigned)p) / (unsigned)a.P.S) - (int)(q0 / (unsigned)a.P.S);      // 0 .. 4
                const double z = *(const double*)(buf + 2 * p);
                const float* ro_ = (const float*)(buf + 64 + 6 * rr);             // lane 6 rr + k of the origin operation
                const float* rd_ = (const float*)(buf + 128 + 6 * rr + 3);        // lane 6 rr + 3 + k of the direction operation
#pragma unroll
                for (int k = 0; k < 3; ++k) pt[k] = __dadd_rn((double)ro_[k], __dmul_rn((double)rd_[k], z));   // load_point's arithmetic
            } else if (a.P.mode == ADFP_PTS_F64) {
#pragma unroll
                for (int k = 0; k < 3; ++k) pt[k] = ((const double*)buf)[3 * p + k];
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) pt[k] = (double)((const float*)buf)[3 * p + k];
            }
        };
        int jc = wvu;
        if (jc < ntl) { dma_small(small, tile_of(jc)); dma_pos(small + SMALL, tile_of(jc)); }
        while (jc < ntl) {
            const int tile = tile_of(jc);
            const int jn = claim();
            const bool more = jn < ntl;
            const int tnext = tile_of(jn);
            const int loc = tile * 32 + p;
            const bool valid = loc < a.total;
            double pt[3];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // nothing else of this role is in the memory queue
            read_small(small, cur);
            read_pos(small + SMALL, tile, pt);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (more) { dma_small(small, tnext); dma_pos(small + SMALL, tnext); }
            float pf[3] = {(float)pt[0], (float)pt[1], (float)pt[2]};
            const bool pnan = (pt[0] != pt[0]) | (pt[1] != pt[1]) | (pt[2] != pt[2]);     // decoded at the origin by the forward
            if (pnan) { pf[0] = 0.f; pf[1] = 0.f; pf[2] = 0.f; }
            const unsigned mw0 = valid ? cur.mw[0] : 0u, mw1 = valid ? cur.mw[1] : 0u, mw2 = valid ? cur.mw[2] : 0u;
            const unsigned mk[5] = {mw0, mw0 >> 16, mw1, mw1 >> 16, mw2};
            float go[4]; f32x16 gh; float sc, isc;
#pragma unroll
            for (int o = 0; o < 4; ++o) go[o] = valid ? cur.go[o] : 0.f;
            head_at(p_wo, go, gh, sc, isc);
            const float ssc = isc * gS;
            if (h == 0) *(f32x4*)(ptab + 4 * p) = f32x4{pf[0], pf[1], pf[2], 0.f};
            const float pxy = h ? pf[1] : pf[0], pz0 = h ? 0.f : pf[2];      // the A operands of the p @ B products below
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
            // ---------------- the Fourier blocks, in the transposed layout: lane = feature 32 b + p, registers = the points kmapH(r, h) ----------------
            // the two d/d pre blocks as the products' left operands stay in registers across the three blocks ...
            f16x8 g0Th[2], g0Tl[2], g3Th[2], g3Tl[2];
            operand(slotS0, g0Th, g0Tl);
            operand(slotS3, g3Th, g3Tl);
            // ... and as chain operands (D layout): split ONCE, then parked as f16 pairs in the slot the block came from (16 words per lane,
            // lane-major 16-byte pieces: conflict-free), read back per Fourier block with four ds_read_b128
            {
                const float back = sc * __uint_as_float((254u - (__float_as_uint(gS) >> 23)) << 23);      // sc / S: S is a power of two
                f16x8 xh[2], xl[2];
                chain_operand(slotS0, back, xh, xl);
                f16x8 yh[2], yl[2];
                chain_operand(slotS3, back, yh, yl);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // every read of the two slots (transposed and D layout) has returned
                park(slotS0, xh, xl);
                park(slotS3, yh, yl);
            }
            int pl = p;                                                       // opaque per tile: what depends on it is recomputed, not hoisted and spilled
            asm volatile("" : "+v"(pl));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const f32x4 bm = *(const f32x4*)(lds + LT::P_BM + (32 * b + pl) * 4);
                const bool real = 32 * b + pl < 93;                           // the three padding features are not inputs
                float cs[16];
                f16x8 eTh[2], eTl[2];
                {
                    // p @ B of the tile's 32 points x this block's 32 features on the f32 matrix pipe: D[point][feature] with A = the
                    // positions (lane = point, k = lane half) and B = embedder._B (lane = feature, k = lane half) lands in the
                    // transposed layout directly (lane = feature, registers = points kmapH(r, h)).  K = 3: (x, y) then (z, 0); each
                    // k step is an f32 fma, so the sum is the forward's fmaf(z, bz, fmaf(y, by, x * bx)).  (On the VALU the same
                    // numbers cost 48 instructions and 16 broadcast ds_read_b128 of the position table per block.)
                    f32x16 argv;
#pragma unroll
                    for (int r = 0; r < 16; ++r) argv[r] = 0.f;
                    argv = __builtin_amdgcn_mfma_f32_32x32x2f32(pxy, h ? bm.y : bm.x, argv, 0, 0, 0);
                    argv = __builtin_amdgcn_mfma_f32_32x32x2f32(pz0, h ? 0.f : bm.z, argv, 0, 0, 0);
                    float e[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float sn, c1;
                        adfp_sincosf(argv[r], sn, c1);
                        e[r] = real ? sn : 0.f;
                        cs[r] = c1;
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
                    unpark(slotS3, xh, xl);
                    mfma_chain_h<2>(ge, ldsu + PW(3, b), lane_off, xh, xl);
                    unpark(slotS0, xh, xl);
                    mfma_chain_h<2>(ge, ldsu + PW(0, b), lane_off, xh, xl);
                }
                write_blk(slotS, ge, ssc);                                    // the point's scale goes in here: in this layout a lane IS the point
                float ga[16];
                read_T(slotS, ga);
                float ex = 0.f, ey = 0.f, ez = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const f32x4 pv = *(const f32x4*)(ptab + 4 * kmapH(r, h));
                    const float g = ga[r] * cs[r];
                    ex = fmaf(g, pv.x, ex); ey = fmaf(g, pv.y, ey); ez = fmaf(g, pv.z, ez);
                }
                eb[b][0] += ex; eb[b][1] += ey; eb[b][2] += ez;
                __builtin_amdgcn_sched_barrier(0);                            // one Fourier block at a time: hoisting the next block's sines above this
            }                                                                 // block's tail keeps 32 more registers alive, and they do not exist
            jc = jn;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef ADFP_STAMPS_ROLES
        t_loop_ = wall_clock64();
#endif
        __syncthreads();
        put(0, acc[0]); put(1, acc[1]); put(2, acc[2]);
        __syncthreads();
        rows([&](int b, int r, float v) { if (p < (b < 2 ? 32 : 29)) part[F::F_PL(0) + 32 * b + kmapH(r, h) * 93 + p] = v; });      // 93 features
        __syncthreads();
        put(0, acc[3]); put(1, acc[4]); put(2, acc[5]);
        __syncthreads();
        rows([&](int b, int r, float v) { if (p < (b < 2 ? 32 : 29)) part[F::F_PL(3) + 32 * b + kmapH(r, h) * 125 + p] = v; });
        __syncthreads();
        {   // embedder._B [3][93]: nine sums per lane (register 3 b + k = feature block b, coordinate k) as one more block
            f32x16 e;
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = r < 9 ? eb[r / 3][r % 3] : 0.f;
            put(0, e);
        }
        __syncthreads();
        rows([&](int b, int r, float v) {
            if (b != 0 || r >= 9) return;                                     // wave-uniform
            v += __shfl_xor(v, 32);                                           // the two lane halves hold different points
            const int fb = r / 3, kk = r - 3 * fb;
            if (h == 0 && 32 * fb + p < 93) part[F::F_EB + kk * 93 + 32 * fb + p] = v;
        });
    }
    if (!(a.skip && *a.skip)) report_range(a.status, amax, ADFP_STATUS_F16_RANGE_BWD);
#ifdef ADFP_STAMPS_ROLES
    if (NOUT == 4 && threadIdx.x == 0 && blockIdx.x < 256) {
        g_roles_span[4 * blockIdx.x] = (unsigned long long)role; g_roles_span[4 * blockIdx.x + 1] = t_start_;
        g_roles_span[4 * blockIdx.x + 2] = t_loop_; g_roles_span[4 * blockIdx.x + 3] = wall_clock64();
    }
#endif
}

// Which role's workgroups hold element e of the flat gradient (0 = P, 1 = H, 2 = C): the write-out at the end of k_decode_bwd_roles.
template <int CDIM, int NOUT>
ADFP_DEV int role_of_element(int e) {
    using F = DecLayout<CDIM, NOUT>;
    if (e < F::F_EB) return 2;                                   // fc_c weights and biases
    if (e < F::F_PL(0)) return 0;                                // embedder._B
    if (e >= F::F_OW) return 1;                                  // output_linear
    int i = 0;
#pragma unroll
    for (int k = 1; k < 5; ++k) i = e >= F::F_PL(k) ? k : i;
    const int off = e - F::F_PL(i), ind = F::in_dim(i);
    if (off >= 32 * ind) return 1;                               // a pts_linears bias
    if (i == 0) return 0;
    if (i == 3) return off % ind < 93 ? 0 : 1;                   // layer 3: the Fourier columns, then the h_2 columns
    return 1;
}
// flat[e] += 2^-k sum over the slots of e's role of partial[slot][e]: k_reduce_partials_scaled for the slots k_decode_bwd_roles
// wrote (slots [0, nP) = role P, [nP, nP + nH) = H, the rest = C; elements a role does not own are NOT written there).
template <int CDIM, int NOUT>
__global__ __launch_bounds__(256) void k_reduce_partials_roles(const float* __restrict__ partial, int nP, int nH, int nslots, int stride,
                                                               float* __restrict__ flat, const float* __restrict__ gmax) {
    constexpr int n = DecLayout<CDIM, NOUT>::F_TOTAL;
    __shared__ float s_p[8][32];
    const int ex = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + ex;
    float s0 = 0.f, s1 = 0.f;
    if (e < n) {
        const int role = role_of_element<CDIM, NOUT>(e);
        const int lo = role == 0 ? 0 : (role == 1 ? nP : nP + nH), hi = role == 0 ? nP : (role == 1 ? nP + nH : nslots);
        int k = lo + sg;
        for (; k + 8 < hi; k += 16) { s0 += partial[(long long)k * stride + e]; s1 += partial[(long long)(k + 8) * stride + e]; }
        if (k < hi) s0 += partial[(long long)k * stride + e];
    }
    s_p[sg][ex] = s0 + s1;
    __syncthreads();
    if (sg == 0 && e < n) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += s_p[g][ex];
        const float S = grad_scale(gmax);
        const float inv = __uint_as_float((254u - (__float_as_uint(S) >> 23)) << 23);      // exact reciprocal of a power of two
        flat[e] += t * inv;
    }
}
