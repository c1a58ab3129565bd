// adfp_backward_fused.h -- decoder backward WITH its weight gradients in one kernel: no staging rows, no k_outer_h.
//
// The two-kernel path (adfp_backward_h.h) writes, per point, 1 152 B of cotangent blocks (the G piece) that k_outer_h reads
// back together with the forward's 896 B of layer inputs, re-lays both through LDS into MFMA operands and multiplies them:
// 0.5 ms of a 1.2 ms Mapper iteration for the colour decoder, and -- measured with timing-only variants
// (tools/experiments/build_outer_variants.sh) -- NOT bound by those bytes: with every tile served from cache k_outer_h is 5 %
// faster, with the math removed 5x; it is bound by turning f32 rows into k-major f16 operands on the VALU.
//
// Here the wave that runs a tile's cotangent chain also accumulates the tile's share of every weight gradient:
//
//   * a weight gradient is  dW[u][j] = sum_p G[u][p] X[j][p]  -- a contraction over POINTS, while the chain's registers hold a block as
//     [unit registers][point lanes].  Every block therefore passes through a 4 KB LDS slot once: written as the D-layout pieces
//     (one ds_write_b128 per 4 units, the point's lane slot rotated by 2 q4 + h so that ...) and read back with lane = unit,
//     registers = the 32 points in the order kmapH(r, h) -- 16 conflict-free ds_read_b32.  Split into hi / lo halves the 16
//     registers of a lane ARE the two k-steps of an A or B operand whose k runs over the tile's points, in the same point order
//     for every block.  (A first version transposed by MFMA against a 0/1 matrix: exact, but 4 MFMAs + 32 accumulator reads +
//     16 conversions per block, 21 blocks per tile -- 0.30 ms for the colour decoder; measured, replaced.)
//   * the 15 main 32 x 32 products of a decoder (pts_linears[i] against its input blocks: 3 Fourier + 0, 1, 1, 3 Fourier + 1, 1;
//     fc_c[i] against the grid features: 5) accumulate in 15 x 16 = 240 accumulation registers that live across ALL tiles of
//     the wave -- which is why the workgroup is 256 threads, ONE wave per SIMD: 512 registers per lane, the accumulators in the
//     AGPR half.  Every narrow product (10 biases, embedder._B [3][93], output_linear weight and bias) lands in ONE more accumulator
//     block whose 32 columns are handed out as slots: its B operand carries the narrow factor (ones / x, y, z / d out) in the
//     lanes of its slots and zeros elsewhere.
//   * layer inputs: the forward's X piece (896 B per point, read once) arrives by LDS-DMA (global_load_lds_dwordx4) straight in the
//     rotated slot format -- the rotation is on the SOURCE row each lane fetches -- through a ring of four slots per wave, every
//     refill issued the moment its slot is consumed and at least two layers before it is needed: at one wave per SIMD a load
//     waited for in place is a microsecond of nothing.  The Fourier features are recomputed directly in the transposed layout
//     (lane = feature, positions broadcast from a 512-byte table; sin and cos from one range reduction).  The G piece is never
//     written.
//   * at the end the workgroup's four waves add their accumulators through LDS into one private copy of the flat gradient
//     (bw.partial, as k_outer_h), reduced and unscaled by k_reduce_partials_scaled.
//
// Arithmetic: every product is the 3-product f16 split of two f32 values (the cotangent blocks carry the global power-of-two
// scale S of adfp_backward.h, the per-point scale of the chain is undone before the split), f32 accumulate over the points of a
// wave in tile order.  Used for the 32-channel decoders (low, colour); the high decoder (64 grid channels: 20 main products, does not
// fit) and every case with the in-kernel scatter keep the two-kernel path.
#pragma once
#include "adfp_backward_h.h"
#ifdef ADFP_STAMPS
__device__ unsigned long long g_phase_fused[8];       // debug build only: wave-cycles per phase of k_decode_bwd_fused, summed over waves
#endif

struct DecodeBwdFArgs {
    PtsDev P; NormDev nb;
    const unsigned* packed_t;  // T image
    const float* g_raw;        // [P,4] cotangent of raw (LOW: .w, COLOR: .xyz)
    const unsigned* masks;     // [rows][2][3] from the training forward
    const float* act;          // [rows][DecStage<32>::NXM] layer inputs from the training forward
    float* gc_out;             // [P][32] d/d c rows for k_scatter_sorted, or NULL (no grid gradient wanted)
    int total;                 // points
    int* status; const float* gmax; const int* skip;
    float* partial; int part_stride;          // one private copy of the flat gradient per workgroup (slot blockIdx.x, overwritten; the host reduces gridDim.x slots)
};

// slots (columns) of the narrow-product accumulator
#define FSLOT_BPL(i) (i)            // bias of pts_linears[i]
#define FSLOT_BFC(i) (5 + (i))      // bias of fc_c[i]
#define FSLOT_EB(b, k) (10 + 3 * (b) + (k))   // embedder._B[k][32 b + row]
#define FSLOT_WO(o) (19 + (o))      // output_linear.weight[o][row]; also the ROW that carries d out_o in the GO block
#define FSLOT_BO 23                 // output_linear.bias[o] at row FSLOT_WO(o)

// the 16 registers of a lane -> the hi / lo operand pairs of the block (2 k-steps each)
// (VALU-only blocks are plain float[16], not f32x16: a 16-register tuple has to be contiguous, and the allocator runs out of
// contiguous runs long before it runs out of registers)
template <bool CHECK = true>
ADFP_DEV void split16v(const float* __restrict__ v, f16x8* __restrict__ xh, f16x8* __restrict__ xl, float& amax) {
    split8<CHECK>(v, xh[0], xl[0], amax);
    split8<CHECK>(v + 8, xh[1], xl[1], amax);
}
// acc += A^T-block x B^T-block over the tile's 32 points, 3-product split
ADFP_DEV void outer_job(f32x16& acc, const f16x8* __restrict__ ah, const f16x8* __restrict__ al, const f16x8* __restrict__ bh, const f16x8* __restrict__ bl) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh[ks], acc, 0, 0, 0);
    }
}
// acc[:, slot] += row sums of the block (B = ones in lane `slot`): a bias gradient
ADFP_DEV void rowsum_job(f32x16& acc, const f16x8* __restrict__ ah, const f16x8* __restrict__ al, int lane_n, int slot) {
    asm volatile("" : "+v"(lane_n));                                  // built here, every time: hoisted out of the tile loop the eleven
                                                                      // operands are 44 registers, which then live in scratch
    const unsigned one2 = lane_n == slot ? 0x3C003C00u : 0u;          // f16 (1, 1)
    const f16x8 ones = __builtin_bit_cast(f16x8, u32x4{one2, one2, one2, one2});
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], ones, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], ones, acc, 0, 0, 0);
    }
}

template <int NOUT, int ROLE>
__global__ __launch_bounds__(256) void k_decode_bwd_fused(DecodeBwdFArgs a) {
    constexpr int CDIM = 32;
    using LT = DecLayoutHT<CDIM, NOUT>;
    using ST = DecStage<CDIM>;
    using F = DecLayout<CDIM, NOUT>;
    static_assert(ROLE == ROLE_LOW || ROLE == ROLE_COLOR, "32-channel decoders only");
    // ONE shared array: the T image, then per wave a ring of four X slots (A, B, C, D), the transposition slot S, the slot S3 that
    // parks layer 3's d/d pre block until the Fourier products at the end of the tile (1 024 words each: held in registers
    // instead, its two operand forms were 32 registers of pressure across the densest part of the tile), and the position
    // table (32 x {x, y, z, scale}).  After the tile loop that region holds the workgroup's sum of its waves'
    // accumulators.
    constexpr int SLOT = 1024, XW = 6 * SLOT + 128;               // words per wave (with the colour decoder's image: 160 KB to the byte)
    static_assert(F::F_TOTAL <= 4 * XW, "the reduction copy must fit the per-wave region");
    __shared__ __attribute__((aligned(16))) unsigned ldsu[LT::P_TOTAL + 4 * XW];
    image_to_lds<256, LT::P_TOTAL / 4>(ldsu, a.packed_t);
    __syncthreads();
    const float* lds = (const float*)ldsu;
    float* s_red = (float*)(ldsu + LT::P_TOTAL);

    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    const int lane_off = h * 128 + p * 4;
    const int wave = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
    const int ntiles = (a.total + 31) >> 5;
    float amax = 0.f;
    const float gS = grad_scale(a.gmax);

    // ---- the slot format.  Piece (q4, hh) of a block = units 8 q4 + 4 hh .. + 3 of all 32 points, 16 bytes per point; point pt
    // sits in lane slot hh * 32 + (pt ^ (2 q4 + hh)) of the piece's 1 KB (q4-major, 256 words per q4).  The XOR makes the
    // transposed read -- 32 lanes = 32 units, one point -- hit 32 different banks: bank = 4 ((pt ^ (2 q4 + hh)) & 7) + (unit & 3).
    const int wvu = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned* xs = ldsu + LT::P_TOTAL + wvu * XW;                 // slots A, B, C, D, S, then the table
    unsigned* slotS = xs + 4 * SLOT;
    unsigned* slotS3 = xs + 5 * SLOT;
    float* ptab = (float*)(xs + 6 * SLOT);
    const unsigned xs_addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)xs;
    // as the READER (lane = unit p, half h = points kmapH(r, h) = (r & 3) + 8 (r >> 2) + 4 h): the XOR only touches the point's low
    // three bits, so register r reads at  rbase[r & 3] + 32 (r >> 2)  words: four per-lane bases, the rest is an immediate
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
    // as the WRITER (lane = point p, half h = units kmapH(r, h)): the block's four pieces, scaled
    auto write_blk = [&](unsigned* slot, const auto& v, float s) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
            *(f32x4*)(slot + q4 * 256 + (h * 32 + (p ^ (2 * q4 + h))) * 4) = f32x4{v[4 * q4] * s, v[4 * q4 + 1] * s, v[4 * q4 + 2] * s, v[4 * q4 + 3] * s};
    };
    // X blocks by LDS-DMA (global_load_lds_dwordx4: lane L's 16 bytes land at base + 16 L).  Lane slot (h, s) fetches the row of point
    // s ^ (2 q4 + h).  Issued from inline asm so that the compiler neither counts them nor fences its own LDS reads behind them;
    // landed = my own s_waitcnt.  Rows beyond the end are fetched from row 0 (finite; their cotangents are zero).
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
    constexpr int COL_C = ST::xm(ST::SC);
    if (wave < ntiles) {                                          // the first tile's c, h_4, h_3, h_2 -> A, B, C, D
        dma_x(0, COL_C, wave); dma_x(1, ST::xm(ST::SH(4)), wave); dma_x(2, ST::xm(ST::SH(3)), wave); dma_x(3, ST::xm(ST::SH(2)), wave);
    }

    // accumulators: 0-2 WP0 x e_b, 3 WP1 x h0, 4 WP2 x h1, 5-7 WP3 x e_b, 8 WP3 x h2, 9 WP4 x h3, 10-14 WC_i x c, 15 narrow products
    f32x16 acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

#ifdef ADFP_STAMPS
    unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = clock64();
#endif
    // the small per-point inputs of a tile (position, mask words, d out) are fetched one tile ahead, too: into the SAME registers,
    // once the five layers of the current tile have consumed them (a second register set spilled)
    struct Small { double pt[3]; unsigned mw[3]; float go[4]; };
    auto fetch_small = [&](int tile_n, Small& sm) {
        const int locn = tile_n * 32 + p;
        const bool ok = locn < a.total;
        const int qn = ok ? locn : 0;
        load_point(a.P, qn, sm.pt);
        const unsigned* mrow = a.masks + ((long long)qn * 2 + h) * 3;
        sm.mw[0] = mrow[0]; sm.mw[1] = mrow[1]; sm.mw[2] = mrow[2];
        if (ROLE == ROLE_LOW) { sm.go[0] = a.g_raw[4ll * qn + 3]; sm.go[1] = 0.f; sm.go[2] = 0.f; }
        else { sm.go[0] = a.g_raw[4ll * qn]; sm.go[1] = a.g_raw[4ll * qn + 1]; sm.go[2] = a.g_raw[4ll * qn + 2]; }
        sm.go[3] = 0.f;
    };
    Small cur;
    if (wave < ntiles) fetch_small(wave, cur);
    for (int tile = wave; tile < ntiles; tile += nwaves) {
        ADFP_PHASE(0);                                           // loop overhead
        const int loc = tile * 32 + p;
        const bool valid = loc < a.total;
        const int q = valid ? loc : 0;
        const bool more = tile + nwaves < ntiles;                // wave-uniform
        const int tnext = tile + nwaves;
        // the ring turns by two slots per tile: c / h_4 / h_3 / h_2 of this tile sit in slots rot, rot + 1, rot + 2, rot + 3 (mod 4)
        const int rot = __builtin_amdgcn_readfirstlane(((tile - wave) / nwaves) & 1) * 2;

        float pf[3] = {(float)cur.pt[0], (float)cur.pt[1], (float)cur.pt[2]};
        const bool pnan = (cur.pt[0] != cur.pt[0]) | (cur.pt[1] != cur.pt[1]) | (cur.pt[2] != cur.pt[2]);     // decoded at the origin by the forward
        if (pnan) { pf[0] = 0.f; pf[1] = 0.f; pf[2] = 0.f; }
        const unsigned mw0 = valid ? cur.mw[0] : 0u, mw1 = valid ? cur.mw[1] : 0u, mw2 = valid ? cur.mw[2] : 0u;
        const unsigned mk[5] = {mw0, mw0 >> 16, mw1, mw1 >> 16, mw2};

        // a block out of a slot as operand halves (lane = unit, k = the tile's points)
        auto operand = [&](const unsigned* slot, f16x8* th, f16x8* tl) {
            float v[16];
            read_T(slot, v);
            split16v(v, th, tl, amax);
        };
        auto operand_x = [&](const unsigned* slot, f16x8* th, f16x8* tl) {      // a layer-input block: the forward range-checked these values
            float v[16];
            read_T(slot, v);
            split16v<false>(v, th, tl, amax);
        };

        // ---------------- cotangent of the decoder output ----------------
        float go[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) go[o] = valid ? cur.go[o] : 0.f;
        f32x16 gh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = 0.f;
#pragma unroll
            for (int o = 0; o < NOUT; ++o) s = fmaf(lds[LT::P_WO + (h * NOUT + o) * 16 + r], go[o], s);
            gh[r] = s;
        }
        // per-point power-of-two scale of the chain (k_decode_bwd_h)
        float sc = 1.f, isc = 1.f;
        {
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
        }
        const float ssc = isc * gS;                              // chain scale -> the global scale S of the summed products
        if (h == 0) *(f32x4*)(ptab + 4 * p) = f32x4{pf[0], pf[1], pf[2], ssc};

        ADFP_PHASE(1);                                           // point, masks, d out, d/d h_4, scale
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // c, h_4, h_3, h_2 of this tile have landed (issued during the previous one)
        ADFP_PHASE(2);                                           // waiting for the DMA
        // ---------------- the grid features: the right-hand side of the five fc_c products ----------------
        f16x8 cTh[2], cTl[2];
        operand_x(xs + rot * SLOT, cTh, cTl);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the reads have returned before the DMA may overwrite the slot
        dma_x(rot, ST::xm(ST::SH(1)), tile);                     // c's slot <- h_1 of THIS tile (needed at layer 2)
        // ---------------- output_linear: d out (x) h_4 and its bias ----------------
        {
            // the GO block: d out_o sits at unit FSLOT_WO(o) (D-layout register r of half h <-> unit kmapH(r, h)), zeros elsewhere
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
            operand_x(xs + (rot + 1) * SLOT, hTh, hTl);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma_x(rot + 1, ST::xm(ST::SH(0)), tile);             // h_4's slot <- h_0 of THIS tile (needed at layer 1)
            outer_job(acc[15], hTh, hTl, gTh, gTl);              // [row = h_4 unit][column FSLOT_WO(o)]
            rowsum_job(acc[15], gTh, gTl, p, FSLOT_BO);          // [row FSLOT_WO(o)][column FSLOT_BO]
        }

        ADFP_PHASE(3);                                           // c, output layer
        f32x16 gc;
#pragma unroll
        for (int r = 0; r < 16; ++r) gc[r] = 0.f;
        f16x8 g0Th[2], g0Tl[2], g0h[2], g0l[2];                  // layer 0's d/d pre: transposed (for its Fourier products) and as chain operand

#pragma unroll
        for (int i = 4; i >= 0; --i) {
            f16x8 xh[2], xl[2], tTh[2], tTl[2];
            // fc_c[i]: weight gradient d/d h_i (x) c, bias, and the chain d/d c += Wc_i^T gh
            write_blk(slotS, gh, ssc);
            split16(gh, xh, xl, amax);
            mfma_chain_h<2>(gc, ldsu + LT::T_WC(i), lane_off, xh, xl);
            operand(slotS, tTh, tTl);
            outer_job(acc[10 + i], tTh, tTl, cTh, cTl);
            rowsum_job(acc[15], tTh, tTl, p, FSLOT_BFC(i));
            // through relu
            float gp[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int keep = ((int)(mk[i] << (16 + r))) >> 31;                   // -1 where unit r was active
                gp[r] = __uint_as_float(__float_as_uint(gh[r]) & (unsigned)keep);
            }
            // pts_linears[i]: bias and the products with its input blocks
            unsigned* gslot = i == 3 ? slotS3 : slotS;           // layer 3's block stays in LDS until the Fourier products
            write_blk(gslot, gp, ssc);
            if (i == 0) split16v(gp, g0h, g0l, amax);
            else split16v(gp, xh, xl, amax);
            f32x16 gn;
            if (i > 0) {                                         // the chain towards layer i - 1, in flight while the slot is read back
#pragma unroll
                for (int r = 0; r < 16; ++r) gn[r] = 0.f;
                mfma_chain_h<2>(gn, ldsu + LT::T_WP(i, i == 3 ? 3 : 0), lane_off, xh, xl);
            }
            if (i == 0) { operand(gslot, g0Th, g0Tl); rowsum_job(acc[15], g0Th, g0Tl, p, FSLOT_BPL(0)); }
            else { operand(gslot, tTh, tTl); rowsum_job(acc[15], tTh, tTl, p, FSLOT_BPL(i)); }
            if (i != 0) {                                        // the h_{i-1} input block (layer 3: h_2 beside the Fourier blocks)
                // ring (at rot = 0): i = 4: slot 2 holds h_3, refilled with the next tile's c; 3: slot 3 holds h_2 -> next h_4;
                // 2: slot 0 holds h_1 -> next h_3; 1: slot 1 holds h_0 -> next h_2 -- which is the next tile's layout at rot = 2
                constexpr int RING[5] = {0, 1, 0, 3, 2};
                const int sl = (RING[i] + rot) & 3;
                if (i <= 2) {                                    // h_1 / h_0 of this tile were requested at its start: at least 8 DMA pieces younger
                    if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                f16x8 hTh[2], hTl[2];
                operand_x(xs + sl * SLOT, hTh, hTl);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (more) dma_x(sl, i == 4 ? COL_C : ST::xm(ST::SH(i == 3 ? 4 : (i == 2 ? 3 : 2))), tnext);
                outer_job(acc[i == 4 ? 9 : (i == 3 ? 8 : 2 + i)], tTh, tTl, hTh, hTl);          // i = 1 -> 3, i = 2 -> 4
            }
            if (i > 0) gh = gn;
        }
        ADFP_PHASE(4);                                           // the five layers
        if (a.gc_out && valid) stage_block_scaled(a.gc_out + 32ll * q, 0, h, gc, isc);
        if (more) fetch_small(tnext, cur);                       // the next tile's inputs, in flight during the Fourier blocks

        // ---------------- the Fourier blocks: layers 0 and 3 against sin(p @ B); embedder._B through cos(p @ B) ----------------
        // computed in the transposed layout: lane = feature 32 b + p, registers = the points kmapH(r, h), positions from the table
        // layer 3's d/d pre block back out of its slot: transposed for the products, and as the chain operand (stored x S / point scale)
        f16x8 g3Th[2], g3Tl[2], g3h[2], g3l[2];
        operand(slotS3, g3Th, g3Tl);
        {
            float t[16];
            const float back = sc * (1.0f / gS);                 // stored = chain value x isc x S; both powers of two
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 v = *(const f32x4*)(slotS3 + q4 * 256 + (h * 32 + (p ^ (2 * q4 + h))) * 4);
                t[4 * q4] = v.x * back; t[4 * q4 + 1] = v.y * back; t[4 * q4 + 2] = v.z * back; t[4 * q4 + 3] = v.w * back;
            }
            split16v<false>(t, g3h, g3l, amax);
        }
        int pl = p;                                              // opaque per tile: what depends on it is recomputed, not hoisted out of the
        asm volatile("" : "+v"(pl));                             // tile loop and spilled (the per-lane constants below are 30 registers)
        // the positions as the B operand of the embedder._B products: lane FSLOT_EB(b, k) carries coordinate k of the 16 points, for
        // all three b at once (masked to one b's lanes below); every other lane zero
        f16x8 pkh[2], pkl[2];
        {
            const int ks3 = pl - FSLOT_EB(0, 0);                 // 0..8 in the slot lanes
            const int kc = ks3 - 3 * (ks3 >= 3) - 3 * (ks3 >= 6);
            const bool inslot = ks3 >= 0 && ks3 < 9;
            const unsigned mx = (inslot && kc == 0) ? ~0u : 0u, my = (inslot && kc == 1) ? ~0u : 0u, mz = (inslot && kc == 2) ? ~0u : 0u;
            float pk[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const f32x4 pv = *(const f32x4*)(ptab + 4 * kmapH(r, h));
                pk[r] = __uint_as_float((__float_as_uint(pv.x) & mx) | (__float_as_uint(pv.y) & my) | (__float_as_uint(pv.z) & mz));
            }
            split16v<false>(pk, pkh, pkl, amax);
        }
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const f32x4 bm = *(const f32x4*)(lds + LT::P_BM + (32 * b + pl) * 4);
            const bool real = 32 * b + pl < 93;                  // the three padding features are not inputs
            float e[16], cs[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const f32x4 pv = *(const f32x4*)(ptab + 4 * kmapH(r, h));
                const float arg = fmaf(pv.z, bm.z, fmaf(pv.y, bm.y, pv.x * bm.x));
                float sn, c1;
                adfp_sincosf(arg, sn, c1);
                e[r] = real ? sn : 0.f;
                cs[r] = c1 * pv.w;                               // cos(p @ B) times the point's scale
            }
            f16x8 eTh[2], eTl[2];
            split16v<false>(e, eTh, eTl, amax);
            outer_job(acc[b], g0Th, g0Tl, eTh, eTl);
            outer_job(acc[5 + b], g3Th, g3Tl, eTh, eTl);
            // d/d (p @ B) = (W0_b^T gp_0 + W3_b^T gp_3) . cos(p @ B)
            f32x16 ge;
#pragma unroll
            for (int r = 0; r < 16; ++r) ge[r] = 0.f;
            mfma_chain_h<2>(ge, ldsu + LT::T_WP(3, b), lane_off, g3h, g3l);
            mfma_chain_h<2>(ge, ldsu + LT::T_WP(0, b), lane_off, g0h, g0l);
            write_blk(slotS, ge, 1.f);
            float ga[16];
            read_T(slotS, ga);
#pragma unroll
            for (int r = 0; r < 16; ++r) ga[r] *= cs[r];
            f16x8 aTh[2], aTl[2], bTh[2], bTl[2];
            split16v(ga, aTh, aTl, amax);
            const bool mine = pl >= FSLOT_EB(b, 0) && pl <= FSLOT_EB(b, 2);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const u32x4 z = {0u, 0u, 0u, 0u};
                bTh[ks] = mine ? pkh[ks] : __builtin_bit_cast(f16x8, z);
                bTl[ks] = mine ? pkl[ks] : __builtin_bit_cast(f16x8, z);
            }
            // [row = feature 32 b + j][column FSLOT_EB(b, k)] += sum_p d/d(p @ B)_j x_k
            outer_job(acc[15], aTh, aTl, bTh, bTl);
        }
        ADFP_PHASE(5);                                           // Fourier blocks
    }
#ifdef ADFP_STAMPS
    if (lane == 0) for (int k = 0; k < 6; ++k) atomicAdd(&g_phase_fused[k], ph_[k]);
#endif
    if (!(a.skip && *a.skip)) report_range(a.status, amax, ADFP_STATUS_F16_RANGE_BWD);

    // ---------------- the waves' accumulators -> the workgroup's copy of the flat gradient ----------------
    // D layout of a product: lane (n, h) register r = [row kmapH(r, h)][column n]
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                             // every wave is done with its X slots: the region becomes s_red
    for (int i = threadIdx.x; i < F::F_TOTAL; i += 256) s_red[i] = 0.f;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int u = kmapH(r, h);
                // pts_linears[i].weight [32][in_dim]: row u, column colbase + p
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    if (32 * b + p < 93) {
                        s_red[F::F_PL(0) + u * 93 + 32 * b + p] += acc[b][r];
                        s_red[F::F_PL(3) + u * 125 + 32 * b + p] += acc[5 + b][r];
                    }
                }
                s_red[F::F_PL(1) + u * 32 + p] += acc[3][r];
                s_red[F::F_PL(2) + u * 32 + p] += acc[4][r];
                s_red[F::F_PL(3) + u * 125 + 93 + p] += acc[8][r];
                s_red[F::F_PL(4) + u * 32 + p] += acc[9][r];
#pragma unroll
                for (int i = 0; i < 5; ++i) s_red[F::F_FC(i) + u * CDIM + p] += acc[10 + i][r];
                // narrow products: column p is a slot
                const float v = acc[15][r];
                if (p < 5) s_red[F::F_PL(p) + 32 * F::in_dim(p) + u] += v;
                else if (p < 10) s_red[F::F_FC(p - 5) + 32 * CDIM + u] += v;
                else if (p < 19) { const int bb = (p - 10) / 3, k = (p - 10) % 3; if (32 * bb + u < 93) s_red[F::F_EB + k * 93 + 32 * bb + u] += v; }
                else if (p < 19 + NOUT) s_red[F::F_OW + (p - 19) * 32 + u] += v;
                else if (p == FSLOT_BO && u >= 19 && u < 19 + NOUT) s_red[F::F_OB + (u - 19)] += v;
            }
        }
        __syncthreads();
    }
    float* part = a.partial + (long long)blockIdx.x * a.part_stride;
    for (int i = threadIdx.x; i < F::F_TOTAL; i += 256) part[i] = s_red[i];       // the slot is this workgroup's alone and written whole: no zero fill before the launch
}
