// ARCHIVED EXPERIMENT (round 5) -- not compiled, not part of the library.
//
// A rewrite of k_outer_h (csrc/adfp_backward_h.h) for the attention network's weight gradients: no f32 copy of the tile in LDS, one
// barrier per 16-row tile instead of three, buffer loads two tiles ahead in two register sets (the ISA was checked: both halves of
// the loop wait with vmcnt(30..16), i.e. the younger set stays in flight; 256 VGPRs, spills outside the loop only).
// It passed tests/test_gpu_grad.py, test_gpu_mapper_iteration.py, test_gpu_graph.py and test_gpu_config3.py (62 tests), and it
// bought nothing: fused Mapper iteration, graph replay, same lease, in-tree = this kernel against -DADFP_EXP_OUTER_OLD = k_outer_h:
//     5 000 x 64:  0.7077 ms against 0.7092 ms        1 000 x 48:  0.3316 ms against 0.3311 ms
// k_outer_h's own timeline (profiles/r05_outer_span.txt, tools/experiments/build_outer_span.sh + outer_span.py): a 16-row tile takes
// 2.5 us whether 256 workgroups run 12-16 tiles or 167 run 2-4 -- the loop does not wait for memory; what a tile costs in BOTH
// kernels is the products phase (8 waves x 28 ds_read_b128 of operands = 229 KB of LDS reads per tile and CU, ~1 800 cycles, beside
// 2 x 21 MFMAs per SIMD) plus, in k_outer_h, the stash and the conversion.  9 us pass before the first tile is done and the
// write-out of the workgroup's copy of the gradient takes 9 us (134 KB each, 34 MB in all; k_reduce_partials_scaled then reads
// them back in 12.7 us): 18 of the launch's 48 us at 5 000 x 64, 18 of 22 us at 1 000 x 48, do not depend on the rows.
// This kernel's own timeline (second half of that file): 1.9 us per tile -- at 5 000 x 64 that is 256 x 53 KB per 1.9 us = 7 TB/s, the
// memory system's rate -- and 11.3 us for the write-out (9.4 in k_outer_h): the launch ends 6 us earlier at 5 000 x 64, 1 us later at
// 1 000 x 48; inside the graph replay the iteration gained the 1.5 us above.
// The next step on this path is therefore fewer bytes and fewer operand reads, not fewer barriers: the products inside
// k_attention_bwd_h (no G piece at all, as k_decode_bwd_roles does for the decoders), or workgroup groups that split the JOBS by
// layer (the attention layers use disjoint column blocks) so that a workgroup's copy of the gradient is a fraction of the whole,
// and operands kept in registers across the jobs that share them (the A block of a layer is read once per job today).
//
// ---------------------------------------------------------------------------------------------
// k_outer_h for rows that are stored whole (the attention network: X piece and G piece of PITCH = NC / 2 columns each, no virtual
// columns), without the f32 copy of the tile in LDS.  An f16 operand is 8 consecutive rows of one column per lane, and lanes that
// take CONSECUTIVE columns of the same row read one coalesced piece of it: lane (i, h) of a wave loads rows 8 h .. 8 h + 7 of the
// columns 2 i, 2 i + 1 of one of its column blocks straight from memory (8 dwordx2: 256 bytes of a row per half wave), splits
// them and writes the hi / lo halves into the operand image -- k_outer_h's [hi|lo][rows 0-7 | 8-15][column][8 halves].  A piece is
// six blocks of 64 columns and one of 32; the 14 blocks are dealt to the 8 waves, two slots each (the lanes i >= 16 of a 32-column
// block and the second slot of the last two waves load something valid and write nothing: every wave runs the same instructions).
// The operand image is double-buffered, so a tile costs ONE barrier (k_outer_h: three -- stash, conversion, products), and the
// loads run TWO tiles ahead of the products in two register sets; every fetch is 16 loads whatever the tile, which lets the
// compiler's s_waitcnt vmcnt leave the younger set in flight.
// Same job table, same partial-sum slots, same epilogue as k_outer_h (which stays for the decoders' staged path).
// ---------------------------------------------------------------------------------------------
template <int NC, int JW>
__global__ __launch_bounds__(512) void k_outer_rows(OuterHArgs b) {
    constexpr int PITCH = NC / 2, NBIG = PITCH / 64, NS = 2;
    static_assert(PITCH % 64 == 32 && 2 * NBIG + 2 <= 2 * OUTER_NW - 2 && 2 * NBIG >= OUTER_NW, "k_outer_rows: the block hand-out below");
    const OuterArgs& a = b.o;
    __shared__ __attribute__((aligned(16))) unsigned st[2][4 * NC * 4];
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int hi = a.chunk_hi;
    if (a.count_ptr) { const int cnt = *a.count_ptr; hi = hi < cnt ? hi : cnt; }
    const int rows = hi - a.chunk_lo;
    const int BR = a.rows_per_wave, G = gridDim.x;
    if ((long long)blockIdx.x * BR >= rows) return;
    int nt = 0;                                            // my tiles: blocks blockIdx.x, + G, ... of BR rows, 16 rows a tile
    for (long long r0 = (long long)blockIdx.x * BR; r0 < rows; r0 += (long long)G * BR) {
        const long long left = rows - r0;
        nt += (int)(((left < BR ? left : BR) + OUTER_RT - 1) / OUTER_RT);
    }
    f32x16 acc[JW];
#pragma unroll
    for (int j = 0; j < JW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    int ca[JW], cb[JW];
#pragma unroll
    for (int j = 0; j < JW; ++j) {
        const int job = wv + OUTER_NW * j;
        ca[j] = job < a.njobs ? a.jobs[job].colA + i : i;
        cb[j] = job < a.njobs ? a.jobs[job].colB + i : i;
    }
    // slot 0 of wave w: 64-column block w (blocks 0 .. NBIG - 1 of X, then of G); slot 1: 64-column block 8 + w while there is one, then
    // the two 32-column blocks, then nothing.  The loads are BUFFER loads: a descriptor per piece and tile on the scalar unit (base =
    // the tile's first row, size = the tile's rows), a 32-bit lane offset that does not depend on the tile, the row of the load as
    // the instruction's immediate -- no per-lane 64-bit address arithmetic (written with pointers, the compiler hoists "piece + lane
    // offset" out of the tile loop and adds the tile's offset on the VALU, two instructions per load) -- and the hardware's range
    // check returns 0 for the rows a tile does not have: the last tile of all needs no second code path, and a fetch beyond the
    // last tile (size 0) moves nothing.
    unsigned voff[NS];
    int dcol[NS];                                          // tile column of the lane's first column, -1 = the lane writes nothing
    bool from_x[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k = wv + OUTER_NW * s;                   // 0 .. 2 NBIG - 1: 64 columns; 2 NBIG, 2 NBIG + 1: 32 columns; beyond: none
        int piece, col, width;
        if (k < 2 * NBIG) { piece = k / NBIG; col = 64 * (k % NBIG); width = 64; }
        else if (k < 2 * NBIG + 2) { piece = k - 2 * NBIG; col = 64 * NBIG; width = 32; }
        else { piece = 0; col = 0; width = 0; }
        from_x[s] = piece == 0;
        voff[s] = (unsigned)(col + 2 * i + 8 * h * PITCH) * 4u;
        dcol[s] = 2 * i < width ? piece * PITCH + col + 2 * i : -1;
    }
    const float* const xbase = b.act + (long long)a.chunk_lo * PITCH;
    int cm = blockIdx.x * BR, cblk = blockIdx.x;           // the next tile to fetch: rows cm .. of block cblk, which ends at cm1
    int cm1 = cm + BR < rows ? cm + BR : rows;
    auto advance = [&]() {
        cm += OUTER_RT;
        if (cm >= cm1) { cblk += G; cm = cblk * BR; cm1 = cm + BR < rows ? cm + BR : rows; }
    };
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    float ld0[NS][16], ld1[NS][16];                        // [slot][8 rows of column 2 i | 8 rows of column 2 i + 1]
    auto fetch = [&](float (&ld)[NS][16], bool any) {
        const int tv = any ? (cm1 - cm < OUTER_RT ? cm1 - cm : OUTER_RT) : 0;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(xbase + (long long)cm * PITCH), 0, tv * PITCH * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(a.stage + (long long)cm * PITCH), 0, tv * PITCH * 4, 0x00020000);
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(from_x[s] ? rx : rg, voff[s] + r * (PITCH * 4), 0, 0);
                ld[s][r] = __uint_as_float(t.x); ld[s][8 + r] = __uint_as_float(t.y);
            }
    };
    float amax = 0.f;
    auto convert = [&](float (&ld)[NS][16], unsigned* stb) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            f16x8 xh[2], xl[2];
            split8(ld[s], xh[0], xl[0], amax);
            split8(ld[s] + 8, xh[1], xl[1], amax);
            if (dcol[s] >= 0) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    *(u32x4*)(stb + ((h * NC) + dcol[s] + c) * 4) = __builtin_bit_cast(u32x4, xh[c]);
                    *(u32x4*)(stb + (((2 + h) * NC) + dcol[s] + c) * 4) = __builtin_bit_cast(u32x4, xl[c]);
                }
            }
        }
    };
    auto products = [&](const unsigned* stb) {
#pragma unroll
        for (int j = 0; j < JW; ++j) {
            const f16x8 ah = __builtin_bit_cast(f16x8, *(const u32x4*)(stb + ((0 + h) * NC + ca[j]) * 4));
            const f16x8 al = __builtin_bit_cast(f16x8, *(const u32x4*)(stb + ((2 + h) * NC + ca[j]) * 4));
            const f16x8 bh = __builtin_bit_cast(f16x8, *(const u32x4*)(stb + ((0 + h) * NC + cb[j]) * 4));
            const f16x8 bl = __builtin_bit_cast(f16x8, *(const u32x4*)(stb + ((2 + h) * NC + cb[j]) * 4));
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[j], 0, 0, 0);
        }
    };
    fetch(ld0, true);
    if (nt > 1) advance();
    fetch(ld1, nt > 1);
    // two tiles per trip and ONE way out of the loop, at its end (with a way out between the halves, the compiler's wait-count pass sees a
    // path into the loop's head on which set 0 holds the youngest loads and drains every load before the first conversion)
    int k = 0;
    for (; k + 1 < nt; k += 2) {
        convert(ld0, st[0]);
        if (k + 2 < nt) advance();
        fetch(ld0, k + 2 < nt);
        __syncthreads();                                   // operand image 0 complete; everyone is past the products on image 1
        products(st[0]);
        convert(ld1, st[1]);
        if (k + 3 < nt) advance();
        fetch(ld1, k + 3 < nt);
        __syncthreads();
        products(st[1]);
    }
    if (k < nt) {
        convert(ld0, st[0]);
        __syncthreads();
        products(st[0]);
    }
    if (!(b.skip && *b.skip)) report_range(b.status, amax, ADFP_STATUS_F16_RANGE_BWD);
    float* part = a.partial + (long long)blockIdx.x * a.part_stride;
#pragma unroll
    for (int j = 0; j < JW; ++j) {
        const int job = wv + OUTER_NW * j;
        if (job < a.njobs) {
            const OuterJob jb = a.jobs[job];
            const int c = i - jb.j0;
            if (c >= 0 && c < jb.nc) {
                float old[16];
                if (b.overwrite) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) old[r] = 0.f;
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const int row = kmapH(r, h); old[r] = part[jb.dst + (row < jb.nr ? row : 0) * jb.rs + c * jb.cs]; }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) { const int row = kmapH(r, h); if (row < jb.nr) part[jb.dst + row * jb.rs + c * jb.cs] = old[r] + acc[j][r]; }
            }
        }
    }
}

