// adfp_backward_h.h -- decoder backward on v_mfma_f32_32x32x16_f16 with the 3-product f32 operand split of adfp_decode_h.h
// (the Mapper's gradient tolerance is 2e-4; the split is fp32-grade, ~1e-6).  Included by adfp_kernels.hip after
// adfp_backward.h.
//
// The exact kernel (k_decode_bwd) recomputes the forward inside the backward, both on f32-input MFMA (1/16 of the f16 rate),
// with the forward and the transposed weights read out of ONE padded f32 image.  An f16 MFMA operand is 8 consecutive k
// values per lane, so the transposed chains need their own packed image (k = out units), and two 64 KB images plus the
// scatter structures do not fit 160 KB of LDS at a useful occupancy.  So nothing is recomputed here:
//
//   training forward  k_decode_h<..., TRAIN = 1>   leaves the ReLU masks (24 B / point) and, for a network whose weight
//                                                  gradients are wanted, its layer inputs (the X part of the staging row)
//   k_decode_bwd_h    cotangent chains  gh -> Wc_i^T gh (d/d c),  gp = mask . gh,  Wp_i^T gp (d/d h_{i-1}, d/d e)  out of
//                     the "T" image, scatter of d/d c into the grid gradient (the shared write-combining scatter), and the
//                     gradient blocks of the staging row (G part) for the weight gradients
//   k_outer_lds2      k_outer_lds with its LDS tile assembled from the two row pieces (X from the forward, G from here)
//
// d/d position (the Tracker's pose gradient) stays on the exact kernel.
#pragma once
#include "adfp_decode_h.h"
#include "adfp_backward.h"

// ---------------------------------------------------------------------------------------------
// "T" image of one decoder: for every layer the transposed blocks  A[row = in unit][k = out unit]  in the k-step format of
// the H image ([hi|lo][h][32 rows][8 halves] = 512 words per k-step, 2 k-steps = the 32 out units): first fc_c[i]^T
// restricted to the OWN grid's 32 channels, then pts_linears[i]^T one block per 32 in units (layer 0: the three Fourier
// blocks; layer 3: three Fourier blocks + h_2; others: h_{i-1}).  k-step ks, lane half h, element j carries out unit
// kmapH(8 ks + j, h) -- register 8 ks + j of the D-layout cotangent, so a chain's result feeds the next chain unmoved.
// ---------------------------------------------------------------------------------------------
template <int CDIM, int NOUT>
struct DecLayoutHT {
    using F = DecLayout<CDIM, NOUT>;
    __host__ __device__ static constexpr int nb(int i) { return i == 0 ? 3 : (i == 3 ? 4 : 1); }
    static constexpr int P_BM = 0;                                   // [96][4] f32 (d/d embedder._B needs cos(p @ B))
    __host__ __device__ static constexpr int T_WC(int i) {
        int o = 384;
        for (int k = 0; k < i; ++k) o += (1 + nb(k)) * 1024;
        return o;
    }
    __host__ __device__ static constexpr int T_WP(int i, int ib) { return T_WC(i) + 1024 * (1 + ib); }
    static constexpr int P_WO = T_WC(5);                              // [2][NOUT][16] f32, as in the H image
    static constexpr int P_TOTAL = P_WO + 2 * NOUT * 16;
};

template <int CDIM, int NOUT>
__device__ HSrc dec_ht_src(int t) {
    using L = DecLayoutHT<CDIM, NOUT>;
    using F = DecLayout<CDIM, NOUT>;
    if (t < 384) {
        const int j = t >> 2, c = t & 3;
        return HSrc{0, (j < 93 && c < 3) ? F::F_EB + c * 93 + j : -1, -1};
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        if (t < L::T_WC(i + 1)) {
            const int u = t - L::T_WC(i);
            const int blk = u >> 10, v = u & 1023;
            const int ks = v >> 9, part = (v >> 8) & 1, h = (v >> 7) & 1, row = (v >> 2) & 31, jp = (v & 3) * 2;
            int src[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int out = kmapH(8 * ks + jp + e, h);
                if (blk == 0) src[e] = F::F_FC(i) + out * CDIM + row;
                else {
                    const int ib = blk - 1;
                    int col;
                    if (i == 0) { col = 32 * ib + row; if (col >= 93) col = -1; }
                    else if (i == 3) { if (ib < 3) { col = 32 * ib + row; if (col >= 93) col = -1; } else col = 93 + row; }
                    else col = row;
                    src[e] = col < 0 ? -1 : F::F_PL(i) + out * F::in_dim(i) + col;
                }
            }
            return HSrc{1 + part, src[0], src[1]};
        }
    }
    const int u = t - L::P_WO;
    const int h = u / (NOUT * 16), o = (u >> 4) % NOUT, r = u & 15;
    return HSrc{0, F::F_OW + o * 32 + kmapH(r, h), -1};
}

template <int CDIM, int NOUT>
__global__ void k_pack_decoder_ht(const float* __restrict__ flat, unsigned* __restrict__ packed, int* __restrict__ status) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= DecLayoutHT<CDIM, NOUT>::P_TOTAL) return;
    const HSrc s = dec_ht_src<CDIM, NOUT>(t);
    if (s.kind == 0) { packed[t] = s.s0 < 0 ? 0u : __float_as_uint(flat[s.s0]); return; }
    const float a = s.s0 < 0 ? 0.f : flat[s.s0], b = s.s1 < 0 ? 0.f : flat[s.s1];
    if (status && !(fmaxf(fabsf(a), fabsf(b)) < 65504.0f))
        __hip_atomic_fetch_or(status, ADFP_STATUS_F16_RANGE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const float ah = f16_hi_part(a), bh = f16_hi_part(b);
    _Float16 x, y;
    if (s.kind == 1) { x = (_Float16)ah; y = (_Float16)bh; }
    else { x = (_Float16)(a - ah); y = (_Float16)(b - bh); }
    packed[t] = (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}

struct DecodeBwdHArgs {
    PtsDev P; NormDev nb;
    GridDev g0;                // own grid (shape only: the scatter's voxel indices)
    const unsigned* packed_t;  // T image
    const int* list; const int* count_ptr;
    const float* g_raw;        // [P,4] cotangent of raw (LOW: .w, COLOR: .xyz)
    const float* att_g;        // HIGH: cotangent per list entry
    const unsigned* masks;     // [rows][2][3] from the training forward (row = point, or list entry for HIGH)
    float* g_grid;             // channels-last gradient of the own grid (or NULL)
    float* stage;              // G part of this chunk's staging rows (WGRAD) or NULL
    int chunk_lo, chunk_hi;
    int* status;
};

// 16 D-layout registers -> the two k-steps of a B operand
ADFP_DEV void split16(const f32x16& v, f16x8* __restrict__ xh, f16x8* __restrict__ xl, float& amax) {
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = v[r];
    split8(t, xh[0], xl[0], amax);
    split8(t + 8, xh[1], xl[1], amax);
}

ADFP_DEV void stage_block_scaled(float* __restrict__ row, int col, int h, const f32x16& v, float s) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 t = {v[4 * q + 0] * s, v[4 * q + 1] * s, v[4 * q + 2] * s, v[4 * q + 3] * s};
        *(f32x4*)(row + col + 8 * q + 4 * h) = t;
    }
}

template <int CDIM, int NOUT, int ROLE, bool WGRAD, int NT>
__global__ __launch_bounds__(NT) void k_decode_bwd_h(DecodeBwdHArgs a) {
    using LT = DecLayoutHT<CDIM, NOUT>;
    using ST = DecStage<CDIM>;
    constexpr int NW = NT / 64;
    constexpr bool CACHE = true;
    __shared__ __attribute__((aligned(16))) unsigned ldsu[LT::P_TOTAL];
    __shared__ float s_tr[NW][32 * 33];
    __shared__ int s_vox[NW][32 * 8];
    __shared__ float s_cw[NW][32 * 8];
    __shared__ float s_cacc[NW][2][32 * 32];
    __shared__ int s_ctag[NW][2][32];
    for (int i = threadIdx.x; i < LT::P_TOTAL / 4; i += NT) ((u32x4*)ldsu)[i] = ((const u32x4*)a.packed_t)[i];
    for (int i = threadIdx.x; i < NW * 2 * 32 * 32; i += NT) (&s_cacc[0][0][0])[i] = 0.f;
    for (int i = threadIdx.x; i < NW * 2 * 32; i += NT) (&s_ctag[0][0][0])[i] = -1;
    __syncthreads();
    const float* lds = (const float*)ldsu;

    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    const ScatterSmem sm = {s_tr[wv], s_vox[wv], s_cw[wv], &s_cacc[wv][0][0], &s_ctag[wv][0][0]};
    const int lane_off = h * 128 + p * 4;
    const int wave = blockIdx.x * NW + wv;
    const int nwaves = gridDim.x * NW;
    int hi = a.chunk_hi;
    if (ROLE == ROLE_HIGH) { const int cnt = *a.count_ptr; hi = hi < cnt ? hi : cnt; }
    const int count = hi - a.chunk_lo;
    const int ntiles = count > 0 ? (count + 31) >> 5 : 0;
    float amax = 0.f;

    for (int tile = wave; tile < ntiles; tile += nwaves) {
        const int loc = tile * 32 + p;
        const bool valid = loc < count;
        const int idx = a.chunk_lo + (valid ? loc : 0);
        const int q = (ROLE == ROLE_HIGH) ? a.list[idx] : idx;
        // G part of the staging row, addressed with the full row's column numbers
        float* srow = WGRAD ? a.stage + (long long)loc * ST::NG - ST::NX : nullptr;

        double pt[3]; float pn[3];
        load_point(a.P, q, pt);
        normalize3(a.nb, pt, pn);

        const unsigned* mrow = a.masks + ((long long)idx * 2 + h) * 3;
        const unsigned mw0 = valid ? mrow[0] : 0u, mw1 = valid ? mrow[1] : 0u, mw2 = valid ? mrow[2] : 0u;
        const unsigned mk[5] = {mw0, mw0 >> 16, mw1, mw1 >> 16, mw2};      // bit 15 - r of the low half: unit r active

        // ---------------- cotangent of the decoder output ----------------
        float go[4] = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
            if (ROLE == ROLE_LOW) go[0] = a.g_raw[4ll * q + 3];
            else if (ROLE == ROLE_COLOR) { go[0] = a.g_raw[4ll * q]; go[1] = a.g_raw[4ll * q + 1]; go[2] = a.g_raw[4ll * q + 2]; }
            else go[0] = a.att_g[idx];
        }
        if (WGRAD && valid) stage_head(srow, ST::SGO, h, f32x4{go[0], go[1], go[2], go[3]});

        // d/d h_4 = Wo^T g_out (VALU, f32)
        f32x16 gh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = 0.f;
#pragma unroll
            for (int o = 0; o < NOUT; ++o) s = fmaf(lds[LT::P_WO + (h * NOUT + o) * 16 + r], go[o], s);
            gh[r] = s;
        }
        // Per-point power-of-two scale.  Cotangents are small numbers (a sample's share of a ray's loss) and shrink further
        // layer by layer; below 6e-5 an f16 half is subnormal and the split's round-toward-zero becomes a relative error
        // (measured 5e-4 on the high decoder's weight gradients without this).  A column of the chain is one point, so every
        // point may carry its own factor: its d/d h_4 is brought to max |.| in [16, 32) -- 11 binades of headroom before the
        // f16 range, and what is lost at the bottom is 2^-29 of the point's largest cotangent.  Undone exactly on the way out.
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
        f32x16 gc;
#pragma unroll
        for (int r = 0; r < 16; ++r) gc[r] = 0.f;

#pragma unroll
        for (int i = 4; i >= 0; --i) {
            if (WGRAD && valid) stage_block_scaled(srow, ST::SGH(i), h, gh, isc);
            f16x8 xh[2], xl[2];
            // through fc_c[i]: d/d c += Wc_i^T gh
            split16(gh, xh, xl, amax);
            mfma_chain_h<2>(gc, ldsu + LT::T_WC(i), lane_off, xh, xl);
            // through relu
            f32x16 gp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int keep = ((int)(mk[i] << (16 + r))) >> 31;                   // v_bfe_i32: -1 where unit r was active
                gp[r] = __uint_as_float(__float_as_uint(gh[r]) & (unsigned)keep);
            }
            if (WGRAD && valid) stage_block_scaled(srow, ST::SGP(i), h, gp, isc);
            if (i == 0 && !WGRAD) break;                                              // layer 0 only feeds d/d e
            split16(gp, xh, xl, amax);                                                // |gp| <= |gh|: already range-checked
            if constexpr (WGRAD) {
                // d/d e reaches the Fourier features through layers 3 and 0.  Layer 3's share waits in the row's SGA columns
                // (raw) until layer 0 adds its own and multiplies by cos(p @ B): d/d (p @ B), what d/d embedder._B needs --
                // 48 accumulators live for two short stretches instead of across the whole chain.
                if (i == 3) {
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        f32x16 ge;
#pragma unroll
                        for (int r = 0; r < 16; ++r) ge[r] = 0.f;
                        mfma_chain_h<2>(ge, ldsu + LT::T_WP(3, b), lane_off, xh, xl);
                        if (valid) stage_block(srow, ST::SGA + 32 * b, h, ge);
                    }
                }
                if (i == 0) {
                    float pf[3] = {(float)pt[0], (float)pt[1], (float)pt[2]};
                    const bool pnan = (pt[0] != pt[0]) | (pt[1] != pt[1]) | (pt[2] != pt[2]);     // decoded at the origin by the forward
                    if (pnan) { pf[0] = 0.f; pf[1] = 0.f; pf[2] = 0.f; }
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        f32x16 ge;
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const f32x4 t = valid ? *(const f32x4*)(srow + ST::SGA + 32 * b + 8 * q4 + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
                            ge[4 * q4 + 0] = t.x; ge[4 * q4 + 1] = t.y; ge[4 * q4 + 2] = t.z; ge[4 * q4 + 3] = t.w;
                        }
                        mfma_chain_h<2>(ge, ldsu + LT::T_WP(0, b), lane_off, xh, xl);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const f32x4 bm = *(const f32x4*)(lds + LT::P_BM + (32 * b + kmapH(r, h)) * 4);
                            const float arg = fmaf(pf[2], bm.z, fmaf(pf[1], bm.y, pf[0] * bm.x));
                            ge[r] = (ge[r] * isc) * __builtin_amdgcn_cosf(adfp_turns(arg));
                        }
                        if (valid) stage_block(srow, ST::SGA + 32 * b, h, ge);
                    }
                }
            }
            if (i > 0) {
                f32x16 gn;
#pragma unroll
                for (int r = 0; r < 16; ++r) gn[r] = 0.f;
                mfma_chain_h<2>(gn, ldsu + LT::T_WP(i, i == 3 ? 3 : 0), lane_off, xh, xl);
                gh = gn;
            }
        }
#ifdef ADFP_EXP_NOSCATTER      // timing experiment only (wrong grid gradients): the kernel without its scatter
        if (a.g_grid && gc[0] == 12345.f) a.g_grid[lane] = gc[1];
#else
        if (a.g_grid) {
#pragma unroll
            for (int r = 0; r < 16; ++r) gc[r] *= isc;
            scatter_tile<CACHE>(a.g_grid, a.g0, pn, valid, gc, lane, sm);
        }
#endif
    }
    if (a.g_grid) scatter_flush<CACHE>(a.g_grid, lane, sm);
    report_range(a.status, amax);
}

// ---------------------------------------------------------------------------------------------
// k_outer_lds with the rows in two pieces: X part (columns [0, nx), row pitch nx, indexed by ABSOLUTE row = chunk_lo + m,
// written by the training forward) and G part (columns [nx, ncols), row pitch ncols - nx, indexed by the row inside the
// chunk).  Each of the 8 waves brings in two of the tile's 16 rows; the job table and everything after the tile is in LDS
// are k_outer_lds's.
// ---------------------------------------------------------------------------------------------
struct Outer2Args { OuterArgs o; const float* act; int nx; };
__global__ __launch_bounds__(512) void k_outer_lds2(Outer2Args b) {
    const OuterArgs& a = b.o;
    __shared__ __attribute__((aligned(16))) float sm[2 * OUTER_RT * OUTER_MAXCOLS];
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    int hi = a.chunk_hi;
    if (a.count_ptr) { const int cnt = *a.count_ptr; hi = hi < cnt ? hi : cnt; }
    const int rows = hi - a.chunk_lo;
    const int m0 = blockIdx.x * a.rows_per_wave;          // rows per WORKGROUP
    if (m0 >= rows) return;
    const int m1 = (m0 + a.rows_per_wave < rows) ? m0 + a.rows_per_wave : rows;
    const int nc = a.ncols, nx4 = b.nx / 4, ng4 = (nc - b.nx) / 4;
    f32x16 acc[OUTER_JW];
#pragma unroll
    for (int j = 0; j < OUTER_JW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    int ca[OUTER_JW], cb[OUTER_JW];
#pragma unroll
    for (int j = 0; j < OUTER_JW; ++j) {
        const int job = wv + OUTER_NW * j;
        ca[j] = job < a.njobs ? a.jobs[job].colA + i : -1;
        cb[j] = job < a.njobs ? a.jobs[job].colB + i : -1;
    }
    f32x4 ld[2][4];                                        // this wave's two rows: X pieces lane, lane + 64; G pieces likewise
    auto fetch = [&](int row0) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int m = row0 + 2 * wv + rr;
            const bool ok = m < m1;
            const f32x4* sx = (const f32x4*)(b.act + (long long)(a.chunk_lo + m) * b.nx);
            const f32x4* sg = (const f32x4*)(a.stage + (long long)m * (nc - b.nx));
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int e = lane + 64 * k;
                ld[rr][k] = (ok && e < nx4) ? sx[e] : f32x4{0.f, 0.f, 0.f, 0.f};
                ld[rr][2 + k] = (ok && e < ng4) ? sg[e] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            f32x4* dst = (f32x4*)(sm + buf * OUTER_RT * OUTER_MAXCOLS + (2 * wv + rr) * nc);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int e = lane + 64 * k;
                if (e < nx4) dst[e] = ld[rr][k];
                if (e < ng4) dst[nx4 + e] = ld[rr][2 + k];
            }
        }
    };
    fetch(m0);
    stash(0);
    int buf = 0;
    for (int m = m0; m < m1; m += OUTER_RT, buf ^= 1) {
        const bool more = m + OUTER_RT < m1;
        if (more) fetch(m + OUTER_RT);
        __syncthreads();
        const float* t = sm + buf * OUTER_RT * OUTER_MAXCOLS;
#pragma unroll
        for (int j = 0; j < OUTER_JW; ++j) {
            if (ca[j] >= 0) {
#pragma unroll
                for (int s = 0; s < OUTER_RT / 2; ++s) {
                    const float va = t[(2 * s + h) * nc + ca[j]], vb = t[(2 * s + h) * nc + cb[j]];
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(va, vb, acc[j], 0, 0, 0);
                }
            }
        }
        if (more) stash(buf ^ 1);
    }
    float* part = a.partial + (long long)blockIdx.x * a.part_stride;
#pragma unroll
    for (int j = 0; j < OUTER_JW; ++j) {
        const int job = wv + OUTER_NW * j;
        if (job < a.njobs) {
            const OuterJob jb = a.jobs[job];
            const int c = i - jb.j0;
            if (c >= 0 && c < jb.nc) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = kmapH(r, h);
                    if (row < jb.nr) part[jb.dst + row * jb.rs + c * jb.cs] += acc[j][r];
                }
            }
        }
    }
}
