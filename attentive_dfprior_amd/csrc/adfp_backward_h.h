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
//   k_scatter_sorted  adds the d/d c rows to the grid gradient in spatial order (run-length sums in registers)
//   k_outer_h         weight gradients on f16 MFMA from the two row pieces (X from the forward, G from here)
//
// d/d position (the Tracker's pose gradient: networks and grids frozen) comes from the PGRAD variants of the two chain kernels;
// together with grid or weight gradients (bundle adjustment) it stays on the exact kernel.  The 32-channel decoders' weight
// gradients are formed inside adfp_backward_fused.h's kernel; k_outer_h serves the high decoder and the attention network.
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
ADFP_DEV void pack_decoder_ht_block(int blk, const float* __restrict__ flat, unsigned* __restrict__ packed, int* __restrict__ status, int bit) {
    const int t = blk * 256 + (int)threadIdx.x;
    if (t >= DecLayoutHT<CDIM, NOUT>::P_TOTAL) return;
    const HSrc s = dec_ht_src<CDIM, NOUT>(t);
    if (s.kind == 0) { packed[t] = s.s0 < 0 ? 0u : __float_as_uint(flat[s.s0]); return; }
    float a = s.s0 < 0 ? 0.f : flat[s.s0], b = s.s1 < 0 ? 0.f : flat[s.s1];
    if (status && !(fmaxf(fabsf(a), fabsf(b)) < 65504.0f))      // the same weight trips the forward image too: that network goes exact
        __hip_atomic_fetch_or(status, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    a = f16_clamp(a); b = f16_clamp(b);          // out of range (flagged above / by pack_range_flag): stay finite, 0 x inf must not appear downstream
    const float ah = f16_hi_part(a), bh = f16_hi_part(b);
    _Float16 x, y;
    if (s.kind == 1) { x = (_Float16)ah; y = (_Float16)bh; }
    else { x = (_Float16)(a - ah); y = (_Float16)(b - bh); }
    packed[t] = (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
template <int CDIM, int NOUT>
__global__ void k_pack_decoder_ht(const float* __restrict__ flat, unsigned* __restrict__ packed, int* __restrict__ status, int bit) { pack_decoder_ht_block<CDIM, NOUT>((int)blockIdx.x, flat, packed, status, bit); }

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
    const float* gmax;         // see grad_scale (WGRAD)
    float* gc_out;             // SCAT = false: [P][32] d/d c rows (row = point) for k_scatter_sorted, or NULL
    const int* skip;           // device flag: the forward call was repaired (zero cotangents, state not valid): report nothing
    float* g_pts;              // PGRAD: [P,3] d/d sample position, accumulated (g0.data = the own grid, channels-last, is read then)
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

// SCAT = true: d/d c goes into the grid gradient from inside the kernel (the write-combining scatter of adfp_backward.h; the
// scatter structures limit the workgroup to 6 waves).  SCAT = false: the kernel only writes d/d c, one 128-B row per point
// (a.gc_out), and k_scatter_sorted adds the rows to the grid gradient in spatial order -- the path the host takes whenever the
// grid fits the binning (run_decode_bwd_h).
// PGRAD (the Tracker: pose gradients, networks and grids frozen -- only with WGRAD = SCAT = false): d/d position through the
// Fourier features (the d/d e blocks of layers 3 and 0 stay in 48 accumulators; cos(p @ B) and the 3-vector product on the VALU
// in f32) and through the trilinear lookup of the own grid (d/d c against the eight corner rows, as the exact kernel does).
// The body of a workgroup: `blk` of `nblk` workgroups of ITS launch share (k_decode_bwd_h: the launch; k_decode_bwd_h_pgrad3: one
// decoder's part of it), the T image loaded into `ldsu` (at least DecLayoutHT<CDIM, NOUT>::P_TOTAL words), `sm` = the wave's scatter
// scratch (SCAT only).  STORE_PTS: d/d position is WRITTEN to a.g_pts (a buffer of this decoder's own: several decoders run side by
// side and k_rays_grad adds the buffers up in a fixed order) instead of added to it.
template <int CDIM, int NOUT, int ROLE, bool WGRAD, bool SCAT, int NT, bool PGRAD, bool STORE_PTS>
ADFP_DEV void decode_bwd_h_body(const DecodeBwdHArgs& a, unsigned* __restrict__ ldsu, const ScatterSmem& sm, int blk, int nblk) {
    static_assert(!PGRAD || (!WGRAD && !SCAT), "the position gradient comes without weight / grid gradients");
    using LT = DecLayoutHT<CDIM, NOUT>;
    using ST = DecStage<CDIM>;
    constexpr int NW = NT / 64;
    constexpr bool CACHE = true;
    image_to_lds<NT, LT::P_TOTAL / 4>(ldsu, a.packed_t);
    __syncthreads();
    const float* lds = (const float*)ldsu;

    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    const int lane_off = h * 128 + p * 4;
    const int wave = blk * NW + wv;
    const int nwaves = nblk * NW;
    int hi = a.chunk_hi;
    if (ROLE == ROLE_HIGH) { const int cnt = *a.count_ptr; hi = hi < cnt ? hi : cnt; }
    const int count = hi - a.chunk_lo;
    const int ntiles = count > 0 ? (count + 31) >> 5 : 0;
    float amax = 0.f;
    const float gS = WGRAD ? grad_scale(a.gmax) : 1.f;

    for (int tile = wave; tile < ntiles; tile += nwaves) {
        const int loc = tile * 32 + p;
        const bool valid = loc < count;
        const int idx = a.chunk_lo + (valid ? loc : 0);
        const int q = (ROLE == ROLE_HIGH) ? a.list[idx] : idx;
        // G piece of the staging row (d/d h_i, d/d (p @ B), d/d out), addressed with the full row's column numbers
        float* srow = WGRAD ? a.stage + (long long)loc * ST::NGM - ST::SGH(0) : nullptr;

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
        if (WGRAD && valid) stage_head(srow, ST::SGO, h, f32x4{go[0] * gS, go[1] * gS, go[2] * gS, go[3] * gS});

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
        const float ssc = WGRAD ? isc * gS : 1.f;            // staged gradient blocks carry the global scale S
        f32x16 gc;
#pragma unroll
        for (int r = 0; r < 16; ++r) gc[r] = 0.f;
        f32x16 gep[PGRAD ? 3 : 1];                           // PGRAD: d/d e, the three 32-feature blocks
        if constexpr (PGRAD) {
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) gep[b][r] = 0.f;
        }

#pragma unroll
        for (int i = 4; i >= 0; --i) {
            if (WGRAD && valid) stage_block_scaled(srow, ST::SGH(i), h, gh, ssc);
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
            if (i == 0 && !WGRAD && !PGRAD) break;                                    // layer 0 only feeds d/d e
            split16(gp, xh, xl, amax);                                                // |gp| <= |gh|: already range-checked
            if constexpr (PGRAD) {
                if (i == 3 || i == 0) {
#pragma unroll
                    for (int b = 0; b < 3; ++b) mfma_chain_h<2>(gep[b], ldsu + LT::T_WP(i, b), lane_off, xh, xl);
                }
            }
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
                            ge[r] = (ge[r] * ssc) * __builtin_amdgcn_cosf(adfp_turns(arg));
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
        if constexpr (PGRAD) {
            float pf[3] = {(float)pt[0], (float)pt[1], (float)pt[2]};
            const bool pnan = (pt[0] != pt[0]) | (pt[1] != pt[1]) | (pt[2] != pt[2]);         // decoded at the origin by the forward
            if (pnan) { pf[0] = 0.f; pf[1] = 0.f; pf[2] = 0.f; }
            // through the Fourier features: d/dp_k = sum_j B[k][j] cos(p @ B)_j d/d e_j
            float gpos[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const f32x4 bm = *(const f32x4*)(lds + LT::P_BM + (32 * b + kmapH(r, h)) * 4);
                    const float arg = fmaf(pf[2], bm.z, fmaf(pf[1], bm.y, pf[0] * bm.x));
                    const float ga = gep[b][r] * __builtin_amdgcn_cosf(adfp_turns(arg));
                    gpos[0] = fmaf(ga, bm.x, gpos[0]); gpos[1] = fmaf(ga, bm.y, gpos[1]); gpos[2] = fmaf(ga, bm.z, gpos[2]);
                }
            // through the trilinear lookup of the OWN grid (the high decoder's low-grid features are under no_grad in the
            // reference, decoder.py:182-187)
            int xi[2], yi[2], zi[2]; float wx[2], wy[2], wz[2], dcx, dcy, dcz;
            tri_axis_d(pn[0], a.g0.X, (float)(2.0 * a.nb.inv[0]), xi[0], xi[1], wx[0], wx[1], dcx);
            tri_axis_d(pn[1], a.g0.Y, (float)(2.0 * a.nb.inv[1]), yi[0], yi[1], wy[0], wy[1], dcy);
            tri_axis_d(pn[2], a.g0.Z, (float)(2.0 * a.nb.inv[2]), zi[0], zi[1], wz[0], wz[1], dcz);
            float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ka = k & 1, kb = (k >> 1) & 1, kc = k >> 2;
                const long long vox = ((long long)zi[kc] * a.g0.Y + yi[kb]) * a.g0.X + xi[ka];
                const f32x4* src = (const f32x4*)(a.g0.data + vox * 32 + 4 * h);
                float sdot = 0.f;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const f32x4 t = src[2 * v];
                    sdot = fmaf(t.x, gc[4 * v + 0], sdot); sdot = fmaf(t.y, gc[4 * v + 1], sdot);
                    sdot = fmaf(t.z, gc[4 * v + 2], sdot); sdot = fmaf(t.w, gc[4 * v + 3], sdot);
                }
                gx = fmaf(sdot, (ka ? 1.f : -1.f) * wy[kb] * wz[kc], gx);
                gy = fmaf(sdot, (kb ? 1.f : -1.f) * wx[ka] * wz[kc], gy);
                gz = fmaf(sdot, (kc ? 1.f : -1.f) * wx[ka] * wy[kb], gz);
            }
            gpos[0] = fmaf(gx, dcx, gpos[0]); gpos[1] = fmaf(gy, dcy, gpos[1]); gpos[2] = fmaf(gz, dcz, gpos[2]);
#pragma unroll
            for (int k = 0; k < 3; ++k) gpos[k] += __shfl_xor(gpos[k], 32);
            if (valid && h == 0) {
                if constexpr (STORE_PTS) {
                    a.g_pts[3ll * q + 0] = gpos[0] * isc; a.g_pts[3ll * q + 1] = gpos[1] * isc; a.g_pts[3ll * q + 2] = gpos[2] * isc;
                } else {
                    a.g_pts[3ll * q + 0] += gpos[0] * isc; a.g_pts[3ll * q + 1] += gpos[1] * isc; a.g_pts[3ll * q + 2] += gpos[2] * isc;
                }
            }
        } else if constexpr (SCAT) {
            if (a.g_grid) {
#pragma unroll
                for (int r = 0; r < 16; ++r) gc[r] *= isc;
                scatter_tile<CACHE>(a.g_grid, a.g0, pn, valid, gc, lane, sm);
            }
        } else {
            if (a.gc_out && valid) stage_block_scaled(a.gc_out + 32ll * q, 0, h, gc, isc);
        }
    }
    if constexpr (SCAT) { if (a.g_grid) scatter_flush<CACHE>(a.g_grid, lane, sm); }
    if (!(a.skip && *a.skip)) report_range(a.status, amax, ADFP_STATUS_F16_RANGE_BWD);
}
template <int CDIM, int NOUT, int ROLE, bool WGRAD, bool SCAT, int NT, bool PGRAD = false>
__global__ __launch_bounds__(NT) void k_decode_bwd_h(DecodeBwdHArgs a) {
    using LT = DecLayoutHT<CDIM, NOUT>;
    constexpr int NW = NT / 64;
    constexpr int NS = SCAT ? NW : 1;
    __shared__ __attribute__((aligned(16))) unsigned ldsu[LT::P_TOTAL];
    __shared__ float s_tr[NS][SCAT ? 32 * 33 : 1];
    __shared__ int s_vox[NS][SCAT ? 32 * 8 : 1];
    __shared__ float s_cw[NS][SCAT ? 32 * 8 : 1];
    __shared__ float s_cacc[NS][2][SCAT ? 32 * 32 : 1];
    __shared__ int s_ctag[NS][2][SCAT ? 32 : 1];
    if constexpr (SCAT) {                                    // (the body's barrier after its image load covers these)
        for (int i = threadIdx.x; i < NW * 2 * 32 * 32; i += NT) (&s_cacc[0][0][0])[i] = 0.f;
        for (int i = threadIdx.x; i < NW * 2 * 32; i += NT) (&s_ctag[0][0][0])[i] = -1;
    }
    const int wv = threadIdx.x >> 6;
    const ScatterSmem sm = {s_tr[SCAT ? wv : 0], s_vox[SCAT ? wv : 0], s_cw[SCAT ? wv : 0], &s_cacc[SCAT ? wv : 0][0][0], &s_ctag[SCAT ? wv : 0][0][0]};
    decode_bwd_h_body<CDIM, NOUT, ROLE, WGRAD, SCAT, NT, PGRAD, false>(a, ldsu, sm, (int)blockIdx.x, (int)gridDim.x);
}
// The position-gradient backward of SEVERAL frozen decoders in ONE launch (the Tracker: high, low and colour decoder): with a few
// hundred tiles each the three launches were three tile latencies in a row (~13 us each at 200 rays); side by side they are one.
// Workgroups [first[k], first[k + 1]) run job k; every job writes d/d position into a buffer of its own (zeroed by the caller: the
// high decoder covers the in-band points only) and k_rays_grad adds the buffers in job order -- the sums of the launches in a row,
// bit for bit (atomics on one buffer made a graph replay differ from the eager sequence in the last bits).
#define ADFP_PGRAD_MAX_JOBS 3
struct DecodeBwdH3Args { DecodeBwdHArgs j[ADFP_PGRAD_MAX_JOBS]; int role[ADFP_PGRAD_MAX_JOBS]; int first[ADFP_PGRAD_MAX_JOBS + 1]; int n; };
__host__ __device__ constexpr int max3i(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
__global__ __launch_bounds__(512) void k_decode_bwd_h_pgrad3(DecodeBwdH3Args m) {
    __shared__ __attribute__((aligned(16))) unsigned ldsu[max3i(DecLayoutHT<64, 1>::P_TOTAL, DecLayoutHT<32, 1>::P_TOTAL, DecLayoutHT<32, 4>::P_TOTAL)];
    int k = 0;
    while (k + 1 < m.n && (int)blockIdx.x >= m.first[k + 1]) ++k;            // block-uniform
    const int blk = (int)blockIdx.x - m.first[k], nblk = m.first[k + 1] - m.first[k];
    const ScatterSmem sm = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const DecodeBwdHArgs& a = m.j[k];
    if (m.role[k] == ROLE_HIGH) decode_bwd_h_body<64, 1, ROLE_HIGH, false, false, 512, true, true>(a, ldsu, sm, blk, nblk);
    else if (m.role[k] == ROLE_LOW) decode_bwd_h_body<32, 1, ROLE_LOW, false, false, 512, true, true>(a, ldsu, sm, blk, nblk);
    else decode_bwd_h_body<32, 4, ROLE_COLOR, false, false, 512, true, true>(a, ldsu, sm, blk, nblk);
}

// ---------------------------------------------------------------------------------------------
// Grid-gradient scatter in spatial order.  Float atomics to the memory-side L2 run at ~325 G lane-adds/s chip-wide, and the
// in-kernel write-combining cache (32 lines per half wave, points in ray order) still sends 40-60 % of the 8 corner lines per
// point there: 150-190 us per grid for a 5 000-ray x 64-sample iteration, five times what the cotangent chains cost.  A camera
// frustum is a small part of the scene -- 320 000 sample points fall into ~8 000 cells of the finest grid -- so:
//
//   k_bin_keys + radix sort     the points ordered by the cell of the coarsest grid they fall in, then by the finest grid's
//   k_scatter_sorted            every half wave walks 64 consecutive sorted points, lane = channel, and sums the 8 corner
//                               contributions in REGISTERS for as long as the cell (of the grid being scattered) stays the
//                               same; a run ends in 8 line-atomics.  ~40 points share a cell on average: a few per cent of
//                               the atomics, and nothing goes through LDS atomics (ds_add_f32 measured at ~170 cycles per
//                               64-lane instruction here, which made an LDS-accumulator variant slower than the cache).
// ---------------------------------------------------------------------------------------------
#define ADFP_BIN_MAXBITS 8            // cells per axis <= 256
// bits of the sort key: the coarse cell's LINEAR index (x fastest) + 6 bits of fine-cell offset.  (A Morton code spends 3 x the bits
// of the LONGEST axis -- 18 for a 37 x 28 x 21 grid whose cells number 21 756 < 2^15 -- and those three bits are a radix pass: 24-bit
// keys sort in three passes, 21-bit keys in two.  The order of the coarse cells among each other does not matter to the runs.)
__host__ __device__ inline int bin_key_bits(int CX, int CY, int CZ) {
    const long long cells = (long long)CX * CY * CZ;
    int b = 0;
    while ((1ll << b) < cells) ++b;
    return b + 6;
}
struct BinArgs {
    PtsDev P; NormDev nb;
    int CX, CY, CZ;            // dims of the coarsest grid that is scattered
    int RX, RY, RZ;            // dims of the finest one (may be the same grid)
    int* key; int* val;        // out: sort key and point id of every point (the radix sort's input)
    // side job of the LAST workgroup (one more than the points need): fold the call's per-ray maxima into the gradient scale's word --
    // this is the first launch after k_composite_bwd, and a launch of its own for one workgroup's work costs ~5 us inside a graph replay
    const float* max_parts; int max_n; float* max_out;
};
ADFP_DEV int cell_axis(float pn, int size) {                 // i0 of tri_axis
    float c = ((pn + 1.f) / 2.f) * (float)(size - 1);
    c = fminf(fmaxf(c, 0.f), (float)(size - 1));
    const int i0 = (int)floorf(c);
    return i0 < 0 ? 0 : i0;
}
ADFP_DEV void bin_keys_block(const BinArgs& a, int blk) {
    const int q = blk * 256 + threadIdx.x;
    if (q >= a.P.n) return;
    double pt[3]; float pn[3];
    load_point(a.P, q, pt);
    normalize3(a.nb, pt, pn);
    // The lattices of a 0.32 m and a 0.16 m grid are incommensurate (align_corners: cell = extent / (dim - 1)), so no order makes
    // the runs of both exact.  Key = index of the COARSE cell, then the fine cell's offset inside it (2 bits per axis): the
    // coarse grid's runs are exact, and a fine cell is cut only where it straddles a coarse face.  (Sorted by the fine cell alone,
    // the points inside a fine cell alternate between coarse cells: a flush per point, 374 us for the coarse grid.)
    const int dims_c[3] = {a.CX, a.CY, a.CZ}, dims_f[3] = {a.RX, a.RY, a.RZ};
    unsigned cc[3], off = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        cc[k] = (unsigned)cell_axis(pn[k], dims_c[k]);
        const int fc = cell_axis(pn[k], dims_f[k]);
        const int f0 = (int)floorf((float)cc[k] * ((float)(dims_f[k] - 1) / (float)(dims_c[k] - 1)) - 1e-3f);
        int d = fc - (f0 < 0 ? 0 : f0);
        d = d < 0 ? 0 : (d > 3 ? 3 : d);
        off |= (unsigned)d << (2 * k);
    }
    a.key[q] = (int)(((((cc[2] * (unsigned)a.CY) + cc[1]) * (unsigned)a.CX + cc[0]) << 6) | off);
    a.val[q] = q;
}
__global__ __launch_bounds__(256) void k_bin_keys(BinArgs a) {
    if (a.max_parts && blockIdx.x == gridDim.x - 1) { max_fold_block<256>(a.max_parts, a.max_n, a.max_out); return; }
    bin_keys_block(a, (int)blockIdx.x);
}

struct ScatterSortedArgs {
    PtsDev P; NormDev nb;
    GridDev g;                 // the grid whose gradient is scattered (dims)
    const float* gc;           // [P][32] d/d c rows
    float* g_grid;
    const int* perm; int n;    // the points in sorted order
    const unsigned char* flags; unsigned flag_mask;      // HIGH: only points with (flags[q] & flag_mask) carry a row
};
// Half h of wave w owns the PPW / 2 consecutive sorted points [PPW w + PPW / 2 h, + PPW / 2) (phase A: one lane per point computes
// cell and weights of the wave's PPW points; phase B: lane = channel).  A run that a range boundary cuts is added in pieces.  (Longer
// ranges per wave only lengthen the critical path: 128 points 1.90 ms per iteration, 1 024 points 2.15 ms; shorter ones cut more
// runs: in round 3, one launch per grid and no merging, 64 points per wave cost 71 us per grid against 46.  Collecting the pieces in
// records and merging them in a SECOND kernel cost more than the atomics it saved; merging them inside the workgroup, below, did not.)
#ifndef ADFP_SCATTER_PPW
#define ADFP_SCATTER_PPW 64           // sorted points per wave (two halves of 32).  Round 5, with all grids in ONE launch and the runs
                                      // merged inside the workgroup: 64 points per wave 57 us against 64 us for 128 (NW = 2 / 3 / 6 / 8 at
                                      // 64 points: 63 / 58 / 60 / 59 us; 256 points: 103 us) -- the shorter serial walk wins once the
                                      // launch no longer ends in three tails
#endif
#ifndef ADFP_SCATTER_NW
#define ADFP_SCATTER_NW 4             // waves per workgroup: the runs that the eight half-wave ranges of a workgroup cut are merged in LDS
#endif
// Round 3: the cells around the camera receive the first samples of EVERY ray (15 000 points in one coarse cell of a 5 000-ray
// batch), and each of the ~230 half-wave ranges that cell is cut into ended in 8 atomics on the SAME 8 lines -- serialised at
// ~17 ns each, the critical path of the launch (62 us for the coarse grid against 20 us for a masked fine one).  A workgroup is
// now four waves = eight consecutive ranges: a range adds its INTERIOR runs to memory as before, but parks its first and its
// last run (the two that may continue next door) in LDS; after a barrier half a wave walks the sixteen records in order and
// adds up neighbours of one cell before they go to memory -- one set of atomics per cell and workgroup instead of one per range.
// One launch for all grids of a backward call (the points and their order are the same; a job = one grid with its d/d c rows):
// three launches of 625 workgroups each ended in their own tail, and a launch costs ~5 us inside a graph replay.
#define ADFP_SCATTER_MAX_JOBS 3
struct ScatterMultiArgs { ScatterSortedArgs j[ADFP_SCATTER_MAX_JOBS]; int n_jobs, blocks_per_job; };
__global__ __launch_bounds__(64 * ADFP_SCATTER_NW) void k_scatter_sorted(ScatterMultiArgs m) {
    constexpr int PPW = ADFP_SCATTER_PPW, PPH = PPW / 2, NW = ADFP_SCATTER_NW;
    const int job = (int)blockIdx.x / m.blocks_per_job, blk = (int)blockIdx.x - job * m.blocks_per_job;
    const ScatterSortedArgs& a = m.j[job];
    __shared__ int s_q[NW][PPW];
    __shared__ int s_cell[NW][PPW];                   // x0 | y0 << 10 | z0 << 20
    __shared__ __attribute__((aligned(16))) float s_w[NW][PPW][8];     // the 8 corner weights (wx wy) wz, corner k = dx + 2 dy + 4 dz
    __shared__ float s_rec[NW * 4][8][32];            // edge records: [range * 2 + (first | last)][corner][channel]
    __shared__ int s_rcell[NW * 4];
    const int lane = threadIdx.x & 63, ch = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    const int w0 = (blk * NW + wv) * PPW;
    // ---- phase A
#pragma unroll
    for (int b = 0; b < PPW / 64; ++b) {
        const int i = w0 + 64 * b + lane;
        int q = -1, cell = -1;
        float w[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (i < a.n) {
            q = a.perm[i];
            if (a.flags && !(a.flags[q] & a.flag_mask)) q = -1;
        }
        if (q >= 0) {
            double pt[3]; float pn[3];
            load_point(a.P, q, pt);
            normalize3(a.nb, pt, pn);
            int x0, y0, z0, x1, y1, z1;
            tri_axis(pn[0], a.g.X, x0, x1, w[0], w[1]);
            tri_axis(pn[1], a.g.Y, y0, y1, w[2], w[3]);
            tri_axis(pn[2], a.g.Z, z0, z1, w[4], w[5]);
            cell = x0 | (y0 << 10) | (z0 << 20);
        }
        s_q[wv][64 * b + lane] = q; s_cell[wv][64 * b + lane] = cell;
        f32x4 lo4, hi4;
#pragma unroll
        for (int k = 0; k < 4; ++k) { lo4[k] = (w[k & 1] * w[2 + (k >> 1)]) * w[4]; hi4[k] = (w[k & 1] * w[2 + (k >> 1)]) * w[5]; }
        *(f32x4*)&s_w[wv][64 * b + lane][0] = lo4; *(f32x4*)&s_w[wv][64 * b + lane][4] = hi4;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- phase B: a run of points in one cell is summed in registers
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int cur = -1;
    bool first_parked = false;
    const int range = wv * 2 + h;                  // 0 .. 2 NW - 1, in sorted order
    auto to_memory = [&](int cell, const float* v) {
        const int x0 = cell & 1023, y0 = (cell >> 10) & 1023, z0 = cell >> 20;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (v[k] != 0.f) {                     // a far-face corner (clamped onto its neighbour) has weight 0: never set
                const int x = x0 + (k & 1), y = y0 + ((k >> 1) & 1), z = z0 + (k >> 2);
                atomicAdd(a.g_grid + ((long long)(z * a.g.Y + y) * a.g.X + x) * 32 + ch, v[k]);
            }
        }
    };
    auto park = [&](int slot) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s_rec[slot][k][ch] = acc[k];
        if (ch == 0) s_rcell[slot] = cur;
    };
    auto flush = [&]() {                           // a run ends inside the range
        if (cur < 0) return;
        if (!first_parked) { park(range * 2); first_parked = true; }     // the range's first run may continue the previous range's last
        else to_memory(cur, acc);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    };
    if (ch == 0) { s_rcell[range * 2] = -1; s_rcell[range * 2 + 1] = -1; }
    for (int j0 = 0; j0 < PPH; j0 += 16) {
        float gv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {             // the rows of the next 16 points, fetched together
            const int q = s_q[wv][PPH * h + j0 + j];
            gv[j] = q >= 0 ? a.gc[32ll * q + ch] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int pi = PPH * h + j0 + j;
            const float g = gv[j];
            const unsigned long long nz = __ballot(g != 0.f);
            if ((unsigned)(nz >> (32 * h)) == 0u) continue;       // no gradient from this point (a dropped ray, a point outside the band)
            const int cell = s_cell[wv][pi];
            if (cell != cur) { flush(); cur = cell; }
            const f32x4 wl = *(const f32x4*)&s_w[wv][pi][0], wh = *(const f32x4*)&s_w[wv][pi][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[k] += g * wl[k]; acc[4 + k] += g * wh[k]; }
        }
    }
    if (cur >= 0) park(range * 2 + (first_parked ? 1 : 0));       // the range's last run (its only one: the first slot)
    __syncthreads();
    // ---- merge: the sixteen edge records in sorted order, neighbours of one cell added up before they go to memory
    if (threadIdx.x < 32) {
        float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int mc = -1;
        for (int r = 0; r < NW * 4; ++r) {
            const int c = s_rcell[r];
            if (c < 0) continue;
            if (c != mc) {
                if (mc >= 0) to_memory(mc, m);
#pragma unroll
                for (int k = 0; k < 8; ++k) m[k] = 0.f;
                mc = c;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) m[k] += s_rec[r][k][ch];
        }
        if (mc >= 0) to_memory(mc, m);
    }
}

// ---------------------------------------------------------------------------------------------
// Weight gradients on f16 MFMA: k_outer_lds's job table and per-workgroup partial sums, with the 16-row tile turned into
// MFMA operands ONCE per tile instead of being read row by row as f32 operands (8 x v_mfma_f32_32x32x2_f32 = 512 cycles per
// job and tile; here 3 x v_mfma_f32_32x32x16_f16 = 96).  An f16 operand is 8 consecutive k per lane and k = the tile's rows, so
// after the coalesced f32 stash every thread takes whole COLUMNS of the tile (16 rows each), splits them into hi / lo halves
// and writes them k-major -- [hi|lo][rows 0-7 | 8-15][column][8 halves], the k-step layout of the H image, so that an operand
// is one conflict-free ds_read_b128.  The rows come in two stored pieces (X from the training forward, indexed by the absolute
// row; G from k_decode_bwd_h, scaled by S = grad_scale) -- the kernel is bound by reading them, so what can be rebuilt is not
// stored: the 96 Fourier columns are recomputed from x, y, z (3 sines per thread) and d/d pre_i is formed from d/d h_i and the
// forward's ReLU mask words -- or, with act = NULL, as whole rows (the attention network).
// ---------------------------------------------------------------------------------------------
struct OuterHArgs {
    OuterArgs o;
    const float* act;          // X pieces (row pitch 4 nxm4 floats, absolute row index) or NULL = whole rows in o.stage
    int nxm4, ngm4;            // f32x4 pieces per stored X / G piece (at most 128 each)
    int g_dst4;                // f32x4 index in the tile row where the stored G piece starts
    int x_gap_at4, x_gap4;     // the stored X piece skips x_gap4 f32x4 of the tile row after its first x_gap_at4 (decoders: the Fourier block)
    int x_skip4;               // decoders: the first x_skip4 f32x4 of a stored X piece (the head [x, y, z, 1, 0 ...]) are NOT stored -- the training
                               // forward does not write what the backward can rebuild (128 B of the 896-B row) -- and the tile's head
                               // comes from the points themselves, by the forward's arithmetic; 0 = the piece is whole
    PtsDev P; const int* list; // decoders: the points (row -> point: list[row], or row itself without a list)
    const unsigned* masks;     // decoders: the forward's ReLU mask words (absolute row index), NULL = no virtual columns
    const float* bm;           // decoders: [96][4] Fourier matrix rows (the packed image's P_BM block)
    int col_se, col_sgp;       // decoders: first column of the recomputed Fourier block / of the masked d/d pre block
    int* status;
    const int* skip;           // as DecodeBwdHArgs.skip
    int overwrite;             // the call's ONLY chunk: a workgroup writes its sums into its slot instead of adding to it -- no zero fill of the
                               // 256 slots (34 MB for the attention network) before the launch; k_reduce_partials_scaled then reads only the
                               // slots of workgroups that had rows (slots_in_use)
};
// br = 0 in OuterArgs.rows_per_wave / k_reduce_partials_scaled: the EVEN split -- every one of the G workgroups takes one contiguous
// block of ceil(rows / G) rows rounded up to whole tiles (the row count is only on the device: with fixed 64-row blocks dealt
// round-robin the busiest workgroup of the 5 000 x 64 iteration ran 16 tiles where the average is 13)
__host__ __device__ inline int even_block(int rows, int G) {
    int br = ((rows + G - 1) / G + OUTER_RT - 1) / OUTER_RT * OUTER_RT;
    return br < OUTER_RT ? OUTER_RT : br;
}
// how many of the nslot partial-sum slots a k_outer_h launch of nslot workgroups over `rows` rows in blocks of `br` wrote
__host__ __device__ inline int slots_in_use(int rows, int br, int nslot) {
    if (br == 0) br = even_block(rows, nslot);
    const int nb = rows > 0 ? (rows + br - 1) / br : 0;
    return nb < nslot ? nb : nslot;
}
__global__ __launch_bounds__(512) void k_outer_h(OuterHArgs b) {
    const OuterArgs& a = b.o;
    __shared__ __attribute__((aligned(16))) float sf[OUTER_RT * OUTER_MAXCOLS];                 // the tile, f32, row-major
    __shared__ __attribute__((aligned(16))) unsigned st[4 * OUTER_MAXCOLS * 4];                   // the tile as operands
    __shared__ __attribute__((aligned(16))) float s_bm[96 * 4];
    __shared__ unsigned s_mask[OUTER_RT][6];
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    int hi = a.chunk_hi;
    if (a.count_ptr) { const int cnt = *a.count_ptr; hi = hi < cnt ? hi : cnt; }
    const int rows = hi - a.chunk_lo;
    // Blocks of BR rows (a multiple of 16) are dealt round-robin: workgroup b takes blocks b, b + G, ...  A block is the workgroup's
    // whole share -- one contiguous range (measured 25 % faster per tile than interleaved tiles): computed by the host when it knows
    // the row count, here (rows_per_wave = 0: even_block) when the count is only on the device (the in-band list; a share computed
    // from the count's upper bound left half the workgroups idle, fixed 64-row blocks gave the busiest workgroup 16 tiles of 13).
    const int BR = a.rows_per_wave ? a.rows_per_wave : even_block(rows, (int)gridDim.x);
    int blk = blockIdx.x, m = blk * BR;
    if (m >= rows) return;
    int m1 = m + BR < rows ? m + BR : rows;               // end of the current block = row limit of fetch()
    const int nc = a.ncols;
    const bool dec = b.masks != nullptr;
    if (dec) for (int t = threadIdx.x; t < 96 * 4; t += 512) s_bm[t] = b.bm[t];
    f32x16 acc[OUTER_JW];
#pragma unroll
    for (int j = 0; j < OUTER_JW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    int ca[OUTER_JW], cb[OUTER_JW];
#pragma unroll
    for (int j = 0; j < OUTER_JW; ++j) {
        const int job = wv + OUTER_NW * j;
        ca[j] = job < a.njobs ? a.jobs[job].colA + i : i;      // a missing job multiplies columns 0..31 by themselves and is never
        cb[j] = job < a.njobs ? a.jobs[job].colB + i : i;      // written out: no branch in the MFMA phase, its LDS reads overlap
    }
    f32x4 ld[2][4];                                        // this wave's two rows: X pieces lane, lane + 64; G pieces likewise (a piece <= 2 KB)
    unsigned mreg = 0u;                                    // threads 0..95: one mask word of the tile
    f32x4 hreg = {0.f, 0.f, 0.f, 0.f};                     // threads 0..15 (decoders): the head of one row of the tile
    auto fetch = [&](int row0) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int m = row0 + 2 * wv + rr;
            const bool ok = m < m1;
            const f32x4* sx = (const f32x4*)(b.act + (long long)(a.chunk_lo + m) * (b.nxm4 * 4));
            const f32x4* sg = (const f32x4*)(a.stage + (long long)m * (b.ngm4 * 4));
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int e = lane + 64 * k;
                ld[rr][k] = (ok && e < b.nxm4) ? sx[e] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int e = lane + 64 * k;
                ld[rr][2 + k] = (ok && e < b.ngm4) ? sg[e] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (dec && threadIdx.x < OUTER_RT * 6) {
            const int m = row0 + (int)threadIdx.x / 6;
            mreg = m < m1 ? b.masks[(long long)(a.chunk_lo + m) * 6 + threadIdx.x % 6] : 0u;
        }
        if (dec && threadIdx.x < OUTER_RT) {               // p.float() of the row's point, NaN -> 0: what the training forward fed its Fourier features (k_decode_h)
            const int m = row0 + (int)threadIdx.x;
            hreg = f32x4{0.f, 0.f, 0.f, 0.f};
            if (m < m1) {
                int q = a.chunk_lo + m;
                if (b.list) q = b.list[q];
                double pt[3];
                load_point(b.P, q, pt);
                const bool pnan = (pt[0] != pt[0]) | (pt[1] != pt[1]) | (pt[2] != pt[2]);
                hreg = pnan ? f32x4{0.f, 0.f, 0.f, 1.f} : f32x4{(float)pt[0], (float)pt[1], (float)pt[2], 1.f};
            }
        }
    };
    float amax = 0.f;
    fetch(m);
    for (;;) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {                   // registers -> f32 tile (the stored X piece lacks the 96 Fourier columns)
            f32x4* dst = (f32x4*)(sf + (2 * wv + rr) * nc);
#pragma unroll
            for (int k = 0; k < 2; ++k) { const int e = lane + 64 * k; if (e >= b.x_skip4 && e < b.nxm4) dst[e < b.x_gap_at4 ? e : e + b.x_gap4] = ld[rr][k]; }
#pragma unroll
            for (int k = 0; k < 2; ++k) { const int e = lane + 64 * k; if (e < b.ngm4) dst[b.g_dst4 + e] = ld[rr][2 + k]; }
        }
        if (dec && threadIdx.x < OUTER_RT * 6) s_mask[threadIdx.x / 6][threadIdx.x % 6] = mreg;
        if (dec && threadIdx.x < OUTER_RT) {
            f32x4* dst = (f32x4*)(sf + threadIdx.x * nc);
            dst[0] = hreg;
            for (int e = 1; e < b.x_skip4; ++e) dst[e] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();                                   // tile complete; everyone is done with the previous operands
        int nm = m + OUTER_RT, nblk = blk, nm1 = m1;       // the tile after this one: same block, or the first of my next block
        if (nm >= m1) { nblk = blk + gridDim.x; nm = nblk * BR; nm1 = nm + BR < rows ? nm + BR : rows; }
        const bool more = nm < rows;
        const int cur_m1 = m1;
        if (more) { m1 = nm1; fetch(nm); m1 = cur_m1; }    // in flight during the conversion and the MFMAs
        if (dec) {                                         // the 96 Fourier columns of the 16 rows from x, y, z: 3 sines per thread
            for (int e = threadIdx.x; e < OUTER_RT * 96; e += 512) {
                const int r = e / 96, jf = e - 96 * r;
                const f32x4 bmr = *(const f32x4*)(s_bm + jf * 4);
                sf[r * nc + b.col_se + jf] = adfp_sinf(fmaf(sf[r * nc + 2], bmr.z, fmaf(sf[r * nc + 1], bmr.y, sf[r * nc] * bmr.x)));
            }
            __syncthreads();
        }
        for (int c = threadIdx.x; c < nc; c += 512) {      // columns -> k-major hi / lo halves
            float v[16];
            if (dec && c >= b.col_sgp && c < b.col_sgp + 160) {              // d/d pre_i = mask . d/d h_i
                const int u = (c - b.col_sgp) & 31, li = (c - b.col_sgp) >> 5;
                const int word = ((u >> 2) & 1) * 3 + (li >> 1), bit = 15 - ((u & 3) | ((u >> 3) << 2)) + 16 * (li & 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = (s_mask[r][word] >> bit) & 1u ? sf[r * nc + c + 160] : 0.f;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = sf[r * nc + c];
            }
            f16x8 xh, xl;
            split8(v, xh, xl, amax);
            *(u32x4*)(st + ((0 * nc) + c) * 4) = __builtin_bit_cast(u32x4, xh);
            *(u32x4*)(st + ((2 * nc) + c) * 4) = __builtin_bit_cast(u32x4, xl);
            split8(v + 8, xh, xl, amax);
            *(u32x4*)(st + ((1 * nc) + c) * 4) = __builtin_bit_cast(u32x4, xh);
            *(u32x4*)(st + ((3 * nc) + c) * 4) = __builtin_bit_cast(u32x4, xl);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < OUTER_JW; ++j) {
            const f16x8 ah = __builtin_bit_cast(f16x8, *(const u32x4*)(st + ((0 + h) * nc + ca[j]) * 4));
            const f16x8 al = __builtin_bit_cast(f16x8, *(const u32x4*)(st + ((2 + h) * nc + ca[j]) * 4));
            const f16x8 bh = __builtin_bit_cast(f16x8, *(const u32x4*)(st + ((0 + h) * nc + cb[j]) * 4));
            const f16x8 bl = __builtin_bit_cast(f16x8, *(const u32x4*)(st + ((2 + h) * nc + cb[j]) * 4));
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[j], 0, 0, 0);
        }
        if (!more) break;
        m = nm; blk = nblk; m1 = nm1;
    }
    if (!(b.skip && *b.skip)) report_range(b.status, amax, ADFP_STATUS_F16_RANGE_BWD);
    float* part = a.partial + (long long)blockIdx.x * a.part_stride;
#pragma unroll
    for (int j = 0; j < OUTER_JW; ++j) {
        const int job = wv + OUTER_NW * j;
        if (job < a.njobs) {
            const OuterJob jb = a.jobs[job];
            const int c = i - jb.j0;
            if (c >= 0 && c < jb.nc) {                     // read the 16 slots, then write them: a += per element serialises 16 round trips
                float old[16];
                if (b.overwrite) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) old[r] = 0.f;
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const int row = kmapH(r, h); old[r] = part[jb.dst + (row < jb.nr ? row : 0) * jb.rs + c * jb.cs]; }    // unconditional: 16 loads in flight
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) { const int row = kmapH(r, h); if (row < jb.nr) part[jb.dst + row * jb.rs + c * jb.cs] = old[r] + acc[j][r]; }
            }
        }
    }
}

// flat[e] += 2^-k sum over the workgroup slots of partial[slot][e]   (k_reduce_partials with the gradient scale undone).
// A workgroup takes 32 elements; its 8 groups of 32 threads each sum every 8th slot and the groups are added in a fixed order
// through LDS (reproducible).  One thread per element over all 256 slots left the chip at 0.26 waves per SIMD: 22 us per network.
// count_ptr (may be NULL): the slots come from ONE k_outer_h launch in overwrite mode over min(*count_ptr, rows_max) rows in blocks of
// br -- only the slots of workgroups that had rows hold sums.
__global__ __launch_bounds__(256) void k_reduce_partials_scaled(const float* __restrict__ partial, int nslots, int stride, int n,
                                                                float* __restrict__ flat, const float* __restrict__ gmax,
                                                                const int* __restrict__ count_ptr = nullptr, int rows_max = 0, int br = 1) {
    if (count_ptr) { const int cnt = *count_ptr; nslots = slots_in_use(cnt < rows_max ? cnt : rows_max, br, nslots); }
    __shared__ float s_p[8][32];
    const int ex = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + ex;
    // (eight loads in flight per thread: with two, the 34 MB of the attention network's 256 slots came in at 2.7 TB/s, the rate of
    // 2 x 4 B x the resident threads per memory latency)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < n) {
        const float* p = partial + e;
        int k = sg;
        for (; k + 56 < nslots; k += 64) {
            const float v0 = p[(long long)k * stride], v1 = p[(long long)(k + 8) * stride], v2 = p[(long long)(k + 16) * stride], v3 = p[(long long)(k + 24) * stride];
            const float v4 = p[(long long)(k + 32) * stride], v5 = p[(long long)(k + 40) * stride], v6 = p[(long long)(k + 48) * stride], v7 = p[(long long)(k + 56) * stride];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3; s0 += v4; s1 += v5; s2 += v6; s3 += v7;
        }
        for (; k < nslots; k += 8) s0 += p[(long long)k * stride];
    }
    s_p[sg][ex] = (s0 + s1) + (s2 + s3);
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

// =============================================================================================
// attention network (mlp_tsdf) backward on f16 MFMA: the decoder scheme again.  The training forward (k_attention_h<1>) leaves
// the ReLU masks, the softmax weights and -- for the weight gradients -- the layer inputs; the chains run out of a "T" image
// (W3^T, W2^T, W1^T as [in-block][out-block] k-step pairs, 128 KB, the one thing in LDS); nothing is recomputed.  The exact
// kernel (k_attention_bwd) stays for the position gradient.
// =============================================================================================
struct AttLayoutHT {
    using F = AttLayout;
    static constexpr int P_A0 = 0;                               // [64][4] f32 = (w0, w1, b, 0), unit order
    static constexpr int T_W3 = 256;                             // 4 in-blocks x 2 out-blocks
    static constexpr int T_W2 = T_W3 + 8 * 1024;                 // 4 x 4
    static constexpr int T_W1 = T_W2 + 16 * 1024;                // 2 x 4
    static constexpr int P_WO = T_W1 + 8 * 1024;                 // [2 h][2 o][32] f32
    static constexpr int P_TOTAL = P_WO + 128;
};
__device__ HSrc att_ht_src(int t) {
    using L = AttLayoutHT;
    using F = AttLayout;
    if (t < L::T_W3) {
        const int k = t >> 2, c = t & 3;
        return HSrc{0, c < 2 ? F::F_W0 + k * 2 + c : (c == 2 ? F::F_B0 + k : -1), -1};
    }
    auto chain = [](int u, int nob, int base, int ld) {          // block (ib, ob) of W^T: rows = in units of ib, k = out units of ob
        const int blk = u >> 10, v = u & 1023;
        const int ib = blk / nob, ob = blk % nob;
        const int ks = v >> 9, part = (v >> 8) & 1, h = (v >> 7) & 1, row = (v >> 2) & 31, jp = (v & 3) * 2;
        const int o0 = 32 * ob + kmapH(8 * ks + jp, h), o1 = 32 * ob + kmapH(8 * ks + jp + 1, h), in = 32 * ib + row;
        return HSrc{1 + part, base + o0 * ld + in, base + o1 * ld + in};
    };
    if (t < L::T_W2) return chain(t - L::T_W3, 2, F::F_W3, 128);
    if (t < L::T_W1) return chain(t - L::T_W2, 4, F::F_W2, 128);
    if (t < L::P_WO) return chain(t - L::T_W1, 4, F::F_W1, 64);
    const int u = t - L::P_WO, h = u >> 6, o = (u >> 5) & 1, j = u & 31;
    return HSrc{0, F::F_WO + o * 64 + unit_of(j, h), -1};
}
ADFP_DEV void pack_attention_ht_block(int blk, const float* __restrict__ flat, unsigned* __restrict__ packed, int* __restrict__ status) {
    const int t = blk * 256 + (int)threadIdx.x;
    if (t >= AttLayoutHT::P_TOTAL) return;
    const HSrc s = att_ht_src(t);
    if (s.kind == 0) { packed[t] = s.s0 < 0 ? 0u : __float_as_uint(flat[s.s0]); return; }
    float a = flat[s.s0], b = flat[s.s1];
    if (status && !(fmaxf(fabsf(a), fabsf(b)) < 65504.0f))
        __hip_atomic_fetch_or(status, ADFP_STATUS_F16_RANGE_ATT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    a = f16_clamp(a); b = f16_clamp(b);          // out of range (flagged above / by pack_range_flag): stay finite, 0 x inf must not appear downstream
    const float ah = f16_hi_part(a), bh = f16_hi_part(b);
    _Float16 x, y;
    if (s.kind == 1) { x = (_Float16)ah; y = (_Float16)bh; }
    else { x = (_Float16)(a - ah); y = (_Float16)(b - bh); }
    packed[t] = (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__global__ void k_pack_attention_ht(const float* __restrict__ flat, unsigned* __restrict__ packed, int* __restrict__ status) { pack_attention_ht_block((int)blockIdx.x, flat, packed, status); }

struct AttBwdHArgs {
    const unsigned* packed_t; const int* list; const int* count_ptr;
    const float* att_occ; const float* att_u;
    const unsigned* masks;     // [rows][2][7] from k_attention_h<1>
    const float* g_weight;     // [P] cotangent of the attention weight output (or NULL)
    float* g_raw;              // [P,4]: .w read as cotangent of the fused occupancy, then overwritten with d/d(high+low)
    float* att_g;              // per list entry: d/d(high+low) for the HIGH backward
    float* stage;              // G piece of this chunk's staging rows (AttStage columns [416, 832)) or NULL
    int chunk_lo, chunk_hi;
    int* status; const float* gmax;
    const int* skip;           // device flag: non-zero = zero gradients for this call (see k_composite_bwd)
    PtsDev P; NormDev nt; TsdfDev t; float* g_pts;     // PGRAD: d/d position through inv_tsdf = f(trilerp(TSDF)), accumulated
};
// 16 values of a float array -> the two k-steps of a B operand
ADFP_DEV void split16a(const float* __restrict__ v, f16x8* __restrict__ xh, f16x8* __restrict__ xl, float& amax) {
    split8(v, xh[0], xl[0], amax);
    split8(v + 8, xh[1], xl[1], amax);
}
template <bool WGRAD, bool PGRAD = false>
__global__ __launch_bounds__(512) void k_attention_bwd_h(AttBwdHArgs a) {
    using T = AttLayoutHT;
    using ST = AttStage;
    __shared__ __attribute__((aligned(16))) unsigned ldsu[T::P_TOTAL];
    image_to_lds<512, T::P_TOTAL / 4>(ldsu, a.packed_t);
    __syncthreads();
    const float* lds = (const float*)ldsu;
    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
    const int lane_off = h * 128 + p * 4;
    const int wave = blockIdx.x * 8 + (threadIdx.x >> 6), nwaves = gridDim.x * 8;
    const int cnt = *a.count_ptr;
    const int hi = a.chunk_hi < cnt ? a.chunk_hi : cnt;
    const int count = hi - a.chunk_lo;
    const int ntiles = count > 0 ? (count + 31) >> 5 : 0;
    const float gS = WGRAD ? grad_scale(a.gmax) : 1.f;
    float amax = 0.f;
    for (int tile = wave; tile < ntiles; tile += nwaves) {
        const int loc = tile * 32 + p;
        const bool valid = loc < count;
        const int idx = a.chunk_lo + (valid ? loc : 0);
        const int q = a.list[idx];
        float* srow = WGRAD ? a.stage + (long long)loc * 416 - ST::AG0 : nullptr;      // addressed with the full row's columns
        const float occ = a.att_occ[idx], u = a.att_u[idx];
        const unsigned* mrow = a.masks + (long long)idx * ADFP_ATT_MASK_WORDS;
        unsigned mk[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) mk[k] = valid ? mrow[h * 7 + k] : 0u;
        const float a0 = __uint_as_float(mrow[6]), a1 = __uint_as_float(mrow[13]);
        // ---- softmax / blend backward: out = a0 occ + a1 u, w = a1
        const float g_out = valid ? a.g_raw[4ll * q + 3] : 0.f;
        const float g_w = (valid && a.g_weight && !(a.skip && *a.skip)) ? a.g_weight[q] : 0.f;
        const float ga0 = g_out * occ, ga1 = g_out * u + g_w;
        const float dot = a0 * ga0 + a1 * ga1;
        const float gl0 = a0 * (ga0 - dot), gl1 = a1 * (ga1 - dot);
        if (WGRAD && valid) stage_head(srow, ST::AGL, h, f32x4{gl0 * gS, gl1 * gS, 0.f, 0.f});
        // per-point power-of-two scale of the cotangents (see k_decode_bwd_h)
        float sc = 1.f, isc = 1.f;
        {
            const float m = fmaxf(fabsf(gl0), fabsf(gl1));
            if (m > 0.f) {
                int se = 127 + 4 + 127 - (int)((__float_as_uint(m) >> 23) & 0xFFu);
                se = se < 1 ? 1 : (se > 253 ? 253 : se);
                sc = __uint_as_float((unsigned)se << 23);
                isc = __uint_as_float((unsigned)(254 - se) << 23);
            }
        }
        const float ssc = isc * gS, s0 = gl0 * sc, s1 = gl1 * sc;
        // ---- layer 3: d/d pre_3 = mask . (WO^T gl)
        float gp3[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float g = fmaf(lds[T::P_WO + (h * 2 + 0) * 32 + j], s0, lds[T::P_WO + (h * 2 + 1) * 32 + j] * s1);
            const int keep = ((int)(mk[5] << j)) >> 31;
            gp3[j] = __uint_as_float(__float_as_uint(g) & (unsigned)keep);
        }
        if (WGRAD && valid) { stage_block_mul(srow, ST::AG3, h, gp3, 0, ssc); stage_block_mul(srow, ST::AG3 + 32, h, gp3, 16, ssc); }
        f16x8 xh[8], xl[8];
        split16a(gp3, xh, xl, amax); split16a(gp3 + 16, xh + 2, xl + 2, amax);
        // ---- layer 2: d/d pre_2 = mask . (W3^T gp3)
        float gp2[64];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) mfma_chain_h<2>(acc, ldsu + T::T_W3 + (ib * 2 + ob) * 1024, lane_off, xh + 2 * ob, xl + 2 * ob);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int keep = ((int)(mk[3 + (ib >> 1)] << (16 * (ib & 1) + r))) >> 31;
                gp2[16 * ib + r] = __uint_as_float(__float_as_uint(acc[r]) & (unsigned)keep);
            }
            if (WGRAD && valid) stage_block_mul(srow, ST::AG2 + 32 * ib, h, gp2, 16 * ib, ssc);
        }
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) split16a(gp2 + 16 * ob, xh + 2 * ob, xl + 2 * ob, amax);
        // ---- layer 1: d/d pre_1 = mask . (W2^T gp2)
        float gp1[64];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) mfma_chain_h<2>(acc, ldsu + T::T_W2 + (ib * 4 + ob) * 1024, lane_off, xh + 2 * ob, xl + 2 * ob);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int keep = ((int)(mk[1 + (ib >> 1)] << (16 * (ib & 1) + r))) >> 31;
                gp1[16 * ib + r] = __uint_as_float(__float_as_uint(acc[r]) & (unsigned)keep);
            }
            if (WGRAD && valid) stage_block_mul(srow, ST::AG1 + 32 * ib, h, gp1, 16 * ib, ssc);
        }
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) split16a(gp1 + 16 * ob, xh + 2 * ob, xl + 2 * ob, amax);
        // ---- layer 0: d/d pre_0 = mask . (W1^T gp1); d/d occ_in through the 2 -> 64 layer
        float gx = 0.f, gxu = 0.f;
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) mfma_chain_h<2>(acc, ldsu + T::T_W1 + (ib * 4 + ob) * 1024, lane_off, xh + 2 * ob, xl + 2 * ob);
            float g0[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int keep = ((int)(mk[0] << (16 * ib + r))) >> 31;
                g0[r] = __uint_as_float(__float_as_uint(acc[r]) & (unsigned)keep);
                gx = fmaf(lds[T::P_A0 + (32 * ib + kmapH(r, h)) * 4], g0[r], gx);
                if (PGRAD) gxu = fmaf(lds[T::P_A0 + (32 * ib + kmapH(r, h)) * 4 + 1], g0[r], gxu);     // d/d u through layer 0
            }
            if (WGRAD && valid) stage_block_mul(srow, ST::AG0 + 32 * ib, h, g0, 0, ssc);
        }
        gx += __shfl_xor(gx, 32);
        const float g_in = a0 * g_out + gx * isc;
        if (valid && h == 0) { a.att_g[idx] = g_in; a.g_raw[4ll * q + 3] = g_in; }
        if constexpr (PGRAD) {
            gxu += __shfl_xor(gxu, 32);
            const float g_u = a1 * g_out + gxu * isc;
            if (valid && h == 0) {                       // as k_attention_bwd: u = clamp(-0.1 log(1/(s + 1e-8) - 1 + 1e-7), +-100), s = clamp(1 - (t + 1)/2, 0, 1)
                double pt[3]; float pn[3], dn[3], gt[3];
                load_point(a.P, q, pt);
                normalize3(a.nt, pt, pn);
#pragma unroll
                for (int k = 0; k < 3; ++k) dn[k] = (float)(2.0 * a.nt.inv[k]);
                const float tv = trilerp_scalar_grad(a.t, pn, dn, gt);
                const float sr = 1.f - (tv + 1.f) / 2.f;
                const float scl = fminf(fmaxf(sr, 0.f), 1.f);
                const float se = scl + 1e-8f;
                const float vv = (1.f / se) - 1.f + 1e-7f;
                const float ur = -0.1f * logf(vv);
                const float du_dt = (sr > 0.f && sr < 1.f && ur > -100.f && ur < 100.f) ? -0.05f / (vv * se * se) : 0.f;
                const float g_t = g_u * du_dt;
                a.g_pts[3ll * q + 0] += g_t * gt[0]; a.g_pts[3ll * q + 1] += g_t * gt[1]; a.g_pts[3ll * q + 2] += g_t * gt[2];
            }
        }
    }
    if (!(a.skip && *a.skip)) report_range(a.status, amax, ADFP_STATUS_F16_RANGE_BWD);
}
