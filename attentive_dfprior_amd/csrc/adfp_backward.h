// adfp_backward.h -- backward kernels of the render path (Mapper loss -> grids + decoder
// parameters, reference src/Mapper.py:457-473 through autograd).  Included by adfp_kernels.hip.
//
//   composite_bwd_block  d(depth, uncertainty, colour)/d(raw), part of k_backward_head    common.py:234-251
//   k_attention_bwd    mlp_tsdf backward on the in-band list              decoder.py:240-258
//   k_decode_bwd<...>  decoder backward: recompute forward (ReLU masks), transposed MFMA chains
//                      out of the SAME padded LDS image, feature-gradient scatter into the
//                      channels-last grid gradient with 128-B shaped float atomics
//   k_outer            weight gradients: dW = sum_points g (x) input as 32x32 MFMA outer
//                      products over point-major staging rows, atomically added to the flat
//                      (state_dict order) gradient
#pragma once
#include "adfp_device.h"

// Global power-of-two scale of the staged gradient blocks.  k_outer_h multiplies staged gradients by staged activations on f16
// MFMA; a cotangent is a small number and an f16 half below 6e-5 is subnormal, so the G part of the staging rows is written
// multiplied by S = 2^k with k chosen from the largest |cotangent of raw| of the call (*gmax, an atomic max of float bits
// filled by k_composite_bwd / k_evalpts_bwd_prep) such that this maximum lands in [64, 128): ten binades of headroom for what
// the transposed chains add, and a row 2^-20 below the largest still splits into normal halves.  Undone exactly when the
// per-workgroup partial sums are reduced (k_reduce_partials).
ADFP_DEV float grad_scale(const float* gmax) {
    if (!gmax) return 1.f;
    const unsigned e = (__float_as_uint(*gmax) >> 23) & 0xFFu;
    if (e == 0u) return 1.f;
    int se = 127 + 6 + 127 - (int)e;
    se = se < 1 ? 1 : (se > 253 ? 253 : se);
    return __uint_as_float((unsigned)se << 23);
}

// a staged block multiplied by a (power-of-two) factor
template <typename VT>
ADFP_DEV void stage_block_mul(float* __restrict__ row, int col, int h, const VT& v, const int voff, float s) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 t = {v[voff + 4 * q + 0] * s, v[voff + 4 * q + 1] * s, v[voff + 4 * q + 2] * s, v[voff + 4 * q + 3] * s};
        *(f32x4*)(row + col + 8 * q + 4 * h) = t;
    }
}

// ------------------------------------------------------------------------------------------
// composite backward.  One wave per ray, lane = sample, up to 4 chunks of 64 samples.
//   w_s = a_s T_s, T_s = prod_{j<s} f_j, f = 1 - a + 1e-10, a = sigmoid(10 occ)
//   dL/da_s = G_s T_s - (sum_{j>s} G_j w_j) / f_s        (cumprod backward, no zeros: f >= 1e-10)
//   G_s = gD' z_s + gV (z_s - depth)^2 + gC . c_s,  gD' = gD - 2 gV (swz - depth sw)
// ------------------------------------------------------------------------------------------
#define CB_MAXC 4
struct CompositeBwdArgs {
    const float* raw; const double* z; int n_rays, S;
    const double* g_depth; const double* g_var; const float* g_color; float* g_raw;
    const unsigned char* keep; float* gmax; const float* g_weight; const int* skip;
};
// workgroup `blk` of the launch (256 threads: four rays); a device function so that k_backward_head can run it beside other jobs
ADFP_DEV void composite_bwd_block(const CompositeBwdArgs& a, int blk) {
    const float* __restrict__ raw = a.raw; const double* __restrict__ z = a.z; const int n_rays = a.n_rays, S = a.S;
    const double* __restrict__ g_depth = a.g_depth; const double* __restrict__ g_var = a.g_var; const float* __restrict__ g_color = a.g_color;
    float* __restrict__ g_raw = a.g_raw; const unsigned char* __restrict__ keep = a.keep; float* __restrict__ gmax = a.gmax;
    const float* __restrict__ g_weight = a.g_weight; const int* __restrict__ skip = a.skip;
    const int lane = threadIdx.x & 63;
    const int ray = blk * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    // skip: the forward call was repaired by the f32 fallback (adfp_fallback.h) -- its training state is not valid, the whole
    // call returns zero gradients
    if ((keep && !keep[ray]) || (skip && *skip)) {                        // a ray the pre-filter dropped: no gradient, whatever its samples hold (NaN * 0)
        for (int s = lane; s < S; s += 64) *(f32x4*)(g_raw + ((long long)ray * S + s) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (gmax && lane == 0) gmax[ray] = 0.f;
        return;
    }
    const int nc = (S + 63) >> 6;
    float alpha[CB_MAXC], T[CB_MAXC], w[CB_MAXC], f[CB_MAXC];
    f32x4 r[CB_MAXC];
    double zz[CB_MAXC];
    float carry = 1.f;
    double sw = 0.0, swz = 0.0;
#pragma unroll
    for (int c = 0; c < CB_MAXC; ++c) {
        alpha[c] = 0.f; T[c] = 0.f; w[c] = 0.f; f[c] = 1.f; zz[c] = 0.0; r[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nc) {
            const int s = c * 64 + lane;
            const bool ok = s < S;
            if (ok) { r[c] = *(const f32x4*)(raw + ((long long)ray * S + s) * 4); zz[c] = z[(long long)ray * S + s]; }
            alpha[c] = ok ? sigmoidf_(10.f * r[c].w) : 0.f;
            f[c] = ok ? (1.f - alpha[c] + 1e-10f) : 1.f;
            float incl = f[c];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float v = __shfl_up(incl, o);
                if (lane >= o) incl *= v;
            }
            float excl = __shfl_up(incl, 1);
            if (lane == 0) excl = 1.f;
            T[c] = carry * excl;
            w[c] = alpha[c] * T[c];
            carry *= __shfl(incl, 63);
            sw += (double)w[c]; swz += (double)w[c] * zz[c];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sw += __shfl_xor(sw, o); swz += __shfl_xor(swz, o); }
    const double depth = swz;
    const double gD = g_depth ? g_depth[ray] : 0.0;
    const double gV = g_var ? g_var[ray] : 0.0;
    const double gDe = gD - 2.0 * gV * (swz - depth * sw);
    float gc0 = 0.f, gc1 = 0.f, gc2 = 0.f;
    if (g_color) { gc0 = g_color[3 * ray]; gc1 = g_color[3 * ray + 1]; gc2 = g_color[3 * ray + 2]; }
    float suffix = 0.f;          // sum of G_j w_j over later chunks
    float mx = 0.f;              // largest |cotangent of raw| of this ray (the f16-split backward's gradient scale)
#pragma unroll
    for (int c = CB_MAXC - 1; c >= 0; --c) {
        if (c < nc) {
            const int s = c * 64 + lane;
            const bool ok = s < S;
            const double dz = zz[c] - depth;
            const float G = (float)(gDe * zz[c] + gV * dz * dz) + (gc0 * r[c].x + gc1 * r[c].y + gc2 * r[c].z);
            const float gw = ok ? G * w[c] : 0.f;
            // inclusive suffix sum across the wave (lanes above)
            float inc = gw;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float v = __shfl_down(inc, o);
                if (lane + o < 64) inc += v;
            }
            const float R = (inc - gw) + suffix;                  // sum_{j>s} G_j w_j
            suffix += __shfl(inc, 0);
            const float ga = G * T[c] - R / f[c];
            const float gocc = ga * 10.f * alpha[c] * (1.f - alpha[c]);
            if (ok) {
                f32x4 o4 = {w[c] * gc0, w[c] * gc1, w[c] * gc2, gocc};
                *(f32x4*)(g_raw + ((long long)ray * S + s) * 4) = o4;
                mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o4.x), fabsf(o4.y))), fmaxf(fabsf(o4.z), fabsf(o4.w)));
            }
        }
    }
    if (gmax) {                  // per-ray maximum, folded by max_fold_block (k_bin_keys' side job) or k_max_reduce
        // The cotangent of the attention-weight output enters the attention backward beside d/d raw (gl = a (ga - dot), |gl| <=
        // |g_w| / 4): it belongs to the same scale, or a loss that lives mostly on w would push S up until the staged rows leave
        // the f16 range.
        if (g_weight) for (int s = lane; s < S; s += 64) mx = fmaxf(mx, fabsf(g_weight[(long long)ray * S + s]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        // (5 000 atomics on one address cost 60 us; so do 5 000 agent-scope LOOKS at one address before a rare atomic: every access
        // to one L2 line takes its turn, ~10 ns each)
        if (lane == 0) gmax[ray] = mx;
    }
}

__global__ __launch_bounds__(1024) void k_max_reduce(const float* __restrict__ parts, int n, float* __restrict__ out) {
    max_fold_block<1024>(parts, n, out);
}

// ------------------------------------------------------------------------------------------
// derivatives w.r.t. the sample position (camera tracking, src/Tracker.py:112-133): the trilinear
// lookup is differentiable in its coordinates (grid_sample backward, 'border' padding zeroes the
// gradient of clipped coordinates) and so is sin(p @ B).
// ------------------------------------------------------------------------------------------
// tri_axis + d(unnormalised, clipped coordinate)/d(world coordinate); dn = d p_n / d p = 2 / (hi - lo)
ADFP_DEV void tri_axis_d(float pn, int size, float dn, int& i0, int& i1, float& w0, float& w1, float& dc) {
    const float c = ((pn + 1.f) / 2.f) * (float)(size - 1);
    tri_axis(pn, size, i0, i1, w0, w1);
    dc = (c <= 0.f || c >= (float)(size - 1)) ? 0.f : 0.5f * (float)(size - 1) * dn;   // clip_coordinates_set_grad
}

// TSDF value and its gradient w.r.t. the world position
ADFP_DEV float trilerp_scalar_grad(const TsdfDev& t, const float pn[3], const float dn[3], float g[3]) {
    int xi[2], yi[2], zi[2]; float wx[2], wy[2], wz[2], dx, dy, dz;
    tri_axis_d(pn[0], t.X, dn[0], xi[0], xi[1], wx[0], wx[1], dx);
    tri_axis_d(pn[1], t.Y, dn[1], yi[0], yi[1], wy[0], wy[1], dy);
    tri_axis_d(pn[2], t.Z, dn[2], zi[0], zi[1], wz[0], wz[1], dz);
    float o = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int a = k & 1, b = (k >> 1) & 1, c = k >> 2;
        const float v = t.data[zi[c] * t.sZ + yi[b] * t.sY + xi[a] * t.sX];
        o = fmaf(v, (wx[a] * wy[b]) * wz[c], o);
        gx = fmaf(v, (a ? 1.f : -1.f) * wy[b] * wz[c], gx);
        gy = fmaf(v, (b ? 1.f : -1.f) * wx[a] * wz[c], gy);
        gz = fmaf(v, (c ? 1.f : -1.f) * wx[a] * wy[b], gz);
    }
    // a corner clamped onto its neighbour (i1 == i0 at the far face) has weight 0 and a clipped coordinate
    g[0] = gx * dx; g[1] = gy * dy; g[2] = gz * dz;
    return o;
}

// ------------------------------------------------------------------------------------------
// scatter of d/d c (a tile's 32 points x 32 channels, D layout) into the channels-last grid gradient; shared by the exact
// and the f16-split decoder backward.  Per-wave LDS: tr [32][33] (the tile transposed to [point][channel]), vox / cw
// [32][8] corner voxel (| cache slot << 27) and weight, and with CACHE the write-combining cache described in k_decode_bwd:
// cacc [2][32 slots][32 ch], ctag [2][32].
// ------------------------------------------------------------------------------------------
struct ScatterSmem { float* tr; int* vox; float* cw; float* cacc; int* ctag; };
template <bool CACHE>
ADFP_DEV void scatter_tile(float* __restrict__ g_grid, const GridDev& g0, const float pn[3], bool valid, const f32x16& gc, int lane, const ScatterSmem& sm) {
    const int p = lane & 31, h = lane >> 5;
    // corner voxels / weights of every point of the tile -> LDS (written by the h == 0 lanes)
    if (h == 0) {
        int xi[2], yi[2], zi[2]; float wx[2], wy[2], wz[2];
        tri_axis(pn[0], g0.X, xi[0], xi[1], wx[0], wx[1]);
        tri_axis(pn[1], g0.Y, yi[0], yi[1], wy[0], wy[1]);
        tri_axis(pn[2], g0.Z, zi[0], zi[1], wz[0], wz[1]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
            const int x = xi[dx], y = yi[dy], z = zi[dz];
            int vox = (z * g0.Y + y) * g0.X + x;
            if constexpr (CACHE) vox |= ((x & 1) | ((y & 1) << 1) | ((z & 1) << 2) | ((((x >> 1) ^ (y >> 1) ^ (z >> 1)) & 3) << 3)) << 27;
            sm.vox[p * 8 + k] = vox;
            sm.cw[p * 8 + k] = valid ? (wx[dx] * wy[dy]) * wz[dz] : 0.f;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sm.tr[p * 33 + kmapH(r, h)] = gc[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int ch = lane & 31;
    if constexpr (CACHE) {
        float* cacc = sm.cacc + h * 1024;
        int* ctag = sm.ctag + h * 32;
        for (int i = 0; i < 16; ++i) {            // half h sweeps points 16 h .. 16 h + 15 in ray order
            const int pi = 16 * h + i;
            const float g = sm.tr[pi * 33 + ch];
            const unsigned long long nz = __ballot(g != 0.f);
            if ((unsigned)(nz >> (32 * h)) == 0u) continue;                // nothing to add from my half (the branch is per half)
            // The 8 corners of a cell sit in 8 different slots (the slot carries the coordinate parities), so their
            // read-modify-writes are independent: all tags and cells are fetched first, then updated, with ONE
            // wave-level fence per point instead of one per corner.  A corner of weight 0 (a far-face corner clamped onto
            // its neighbour: the same voxel, hence the same slot, as that neighbour) is skipped -- it is the only way two
            // corners of a point can meet in a slot.
            int pv[8], tag[8]; float w[8], old[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { pv[k] = sm.vox[pi * 8 + k]; w[k] = sm.cw[pi * 8 + k]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int slot = (unsigned)pv[k] >> 27;
                tag[k] = ctag[slot];
                old[k] = cacc[slot * 32 + ch];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (w[k] == 0.f) continue;
                const int vox = pv[k] & 0x7ffffff, slot = (unsigned)pv[k] >> 27;
                const float v = g * w[k];
                float* cell = cacc + slot * 32 + ch;
                if (tag[k] == vox) *cell = old[k] + v;
                else {
                    if (tag[k] >= 0 && old[k] != 0.f) atomicAdd(g_grid + (long long)tag[k] * 32 + ch, old[k]);
                    *cell = v;
                    if (ch == 0) ctag[slot] = vox;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    } else {
        for (int pp = 0; pp < 32; pp += 2) {
            const int pi = pp + (lane >> 5);
            const float g = sm.tr[pi * 33 + ch];
            if (__ballot(g != 0.f) == 0ull) continue;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float wgt = sm.cw[pi * 8 + k];
                const float v = g * wgt;
                if (v != 0.f) atomicAdd(g_grid + (long long)sm.vox[pi * 8 + k] * 32 + ch, v);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <bool CACHE>
ADFP_DEV void scatter_flush(float* __restrict__ g_grid, int lane, const ScatterSmem& sm) {
    if constexpr (CACHE) {
        const int ch = lane & 31, h = lane >> 5;
        for (int slot = 0; slot < 32; ++slot) {               // write the cached lines back
            const int tag = sm.ctag[h * 32 + slot];
            const float v = sm.cacc[h * 1024 + slot * 32 + ch];
            if (tag >= 0 && v != 0.f) atomicAdd(g_grid + (long long)tag * 32 + ch, v);
        }
    }
}

// ------------------------------------------------------------------------------------------
// decoder backward
// ------------------------------------------------------------------------------------------
struct DecodeBwdArgs {
    PtsDev P; NormDev nb; double b[6];
    GridDev g0, g1;
    const float* packed;
    const int* list; const int* count_ptr;
    const float* g_raw;        // [P,4] cotangent of raw (LOW: .w, COLOR: .xyz)
    const float* att_g;        // HIGH: cotangent per list entry
    float* g_grid;             // channels-last gradient of the OWN grid (or NULL)
    float* stage;              // staging rows of this chunk (WGRAD) or NULL
    float* g_pts;              // [P,3] d/d sample position, accumulated (PGRAD) or NULL
    int chunk_lo, chunk_hi;    // point / list-entry range handled by this launch
    unsigned* dbg_masks;       // debug export of the recomputed ReLU decisions (adfp_train_state.dbg_masks_*) or NULL
};

// lane row j of a transposed chain -> offset of in-unit j inside its in-block of the image
ADFP_DEV int lane_off_T(int j) { return ((j >> 3) * 2 + ((j >> 2) & 1)) * ADFP_RG + (j & 3); }

template <int CDIM, int NOUT, int ROLE, bool WGRAD, bool PGRAD, int NT>
__global__ __launch_bounds__(NT) void k_decode_bwd(DecodeBwdArgs a) {
    constexpr bool NEED_E = WGRAD || PGRAD;
    using L = DecLayout<CDIM, NOUT>;
    using ST = DecStage<CDIM>;
    constexpr int NW = NT / 64;
    __shared__ __attribute__((aligned(16))) float lds[L::P_TOTAL];
    __shared__ float s_tr[NW][32 * 33];          // g_c transpose: [point][channel]
    __shared__ int s_vox[NW][32 * 8];            // corner voxel index per point (| cache slot << 27 when CACHE)
    __shared__ float s_cw[NW][32 * 8];           // corner weight per point
    // Write-combining cache of grid-gradient lines (one voxel = 32 channels = 128 B), private to each
    // half-wave: 32 direct-mapped slots whose index is built from the voxel's coordinate parities, so the 8
    // corners of a cell never collide.  Float atomics run at ~1.3 TB/s chip-wide and far below that when
    // many adders hit one line (the voxels around the camera receive the first samples of EVERY ray), and a
    // 32-sample tile re-visits the same ~100 voxels 256 times; summing in LDS first and adding a line once per
    // eviction took the scatter from 3.1 ms to well under 1 ms of a 5 000-ray x 64-sample iteration.
    // (The 138 KB image of the high decoder leaves no room for it; that kernel only sees the in-band list.)
    constexpr bool CACHE = ROLE != ROLE_HIGH;
    __shared__ float s_cacc[CACHE ? NW : 1][2][CACHE ? 32 * 32 : 1];
    __shared__ int s_ctag[CACHE ? NW : 1][2][32];
    for (int i = threadIdx.x; i < L::P_TOTAL / 4; i += NT) ((f32x4*)lds)[i] = ((const f32x4*)a.packed)[i];
    if constexpr (CACHE) {
        for (int i = threadIdx.x; i < NW * 2 * 32 * 32; i += NT) (&s_cacc[0][0][0])[i] = 0.f;
        for (int i = threadIdx.x; i < NW * 2 * 32; i += NT) (&s_ctag[0][0][0])[i] = -1;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    const ScatterSmem sm = {s_tr[wv], s_vox[wv], s_cw[wv], &s_cacc[CACHE ? wv : 0][0][0], &s_ctag[CACHE ? wv : 0][0][0]};
    const int lane_off = h * ADFP_RG + p * 4;
    const int loT = lane_off_T(p);
    const int wave = blockIdx.x * NW + wv;
    const int nwaves = gridDim.x * NW;
    int hi = a.chunk_hi;
    if (ROLE == ROLE_HIGH) { const int cnt = *a.count_ptr; hi = hi < cnt ? hi : cnt; }
    const int count = hi - a.chunk_lo;
    const int ntiles = count > 0 ? (count + 31) >> 5 : 0;

    for (int tile = wave; tile < ntiles; tile += nwaves) {
        const int loc = tile * 32 + p;                 // row inside the chunk
        const bool valid = loc < count;
        const int idx = a.chunk_lo + (valid ? loc : 0);
        const int q = (ROLE == ROLE_HIGH) ? a.list[idx] : idx;
        float* srow = WGRAD ? a.stage + (long long)loc * ST::NCOLS : nullptr;

        double pt[3]; float pn[3], pf[3];
        load_point(a.P, q, pt);
        normalize3(a.nb, pt, pn);
        pf[0] = (float)pt[0]; pf[1] = (float)pt[1]; pf[2] = (float)pt[2];
        // a NaN position belongs to a ray the Mapper's pre-filter drops (0/0 in its slab test): its cotangent is zero, and the
        // recomputed activations must not turn 0 * NaN into NaN in the staged rows of the weight gradients
        const bool pnan = (pt[0] != pt[0]) | (pt[1] != pt[1]) | (pt[2] != pt[2]);
        if (pnan) { pf[0] = 0.f; pf[1] = 0.f; pf[2] = 0.f; }

        // ---------------- forward recompute (ReLU masks; stage inputs when WGRAD) ----------------
        float c[L::KSC];
        gather16(a.g0, pn, h, c);
        if (CDIM == 64) gather16(a.g1, pn, h, c + 16);
        float e[L::KSE], ce[NEED_E ? L::KSE : 1];
#pragma unroll
        for (int s = 0; s < L::KSE; ++s) {
            const f32x4 bm = *(const f32x4*)(lds + L::P_BM + (unit_of(s, 0) + 4 * h) * 4);
            const float arg = fmaf(pf[2], bm.z, fmaf(pf[1], bm.y, pf[0] * bm.x));
            if constexpr (NEED_E) adfp_sincosf(arg, e[s], ce[s]); else e[s] = adfp_sinf(arg);
        }
        if (WGRAD && valid) {
            stage_head(srow, ST::SX, h, f32x4{pf[0], pf[1], pf[2], 1.f});
            stage_block(srow, ST::SE, h, e, 0); stage_block(srow, ST::SE + 32, h, e, 16); stage_block(srow, ST::SE + 64, h, e, 32);
            stage_block(srow, ST::SC, h, c, 0);
            if (CDIM == 64) stage_block(srow, ST::SC + 32, h, c, 16);
        }
        unsigned mask[5];
        f32x16 hcur, acc;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            bias_init(acc, lds + L::P_BP(i), h);
            if (i == 0) mfma_chain<L::KSE>(acc, lds + L::P_WP(0), lane_off, e);
            else if (i == 3) {
                mfma_chain<L::KSE>(acc, lds + L::P_WP(3), lane_off, e);
                mfma_chain<16>(acc, lds + L::P_WP(3) + L::chain_floats(L::KSE), lane_off, hcur);
            } else mfma_chain<16>(acc, lds + L::P_WP(i), lane_off, hcur);
            mask[i] = pos_mask(acc);
            relu_bias(acc, lds + L::P_BC(i), h);
            mfma_chain<L::KSC>(acc, lds + L::P_WC(i), lane_off, c);
            hcur = acc;
            if (WGRAD && valid) stage_block(srow, ST::SH(i), h, hcur);
        }
        if (a.dbg_masks && valid) {     // the training forward's layout (k_decode_h<TRAIN>): register r of layer i at bit 15 - r of its 16-bit field
            unsigned* mrow = a.dbg_masks + ((long long)idx * 2 + h) * 3;
            mrow[0] = (__brev(mask[0]) >> 16) | (__brev(mask[1]) & 0xFFFF0000u);
            mrow[1] = (__brev(mask[2]) >> 16) | (__brev(mask[3]) & 0xFFFF0000u);
            mrow[2] = __brev(mask[4]) >> 16;
        }

        // ---------------- cotangent of the decoder output ----------------
        float go[4] = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
            if (ROLE == ROLE_LOW) go[0] = a.g_raw[4ll * q + 3];
            else if (ROLE == ROLE_COLOR) { go[0] = a.g_raw[4ll * q]; go[1] = a.g_raw[4ll * q + 1]; go[2] = a.g_raw[4ll * q + 2]; }
            else go[0] = a.att_g[idx];
        }
        if (WGRAD && valid) stage_head(srow, ST::SGO, h, f32x4{go[0], go[1], go[2], go[3]});

        // d/d h_4 = Wo^T g_out
        f32x16 gh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = 0.f;
#pragma unroll
            for (int o = 0; o < NOUT; ++o) s = fmaf(lds[L::P_WO + (h * NOUT + o) * 16 + r], go[o], s);
            gh[r] = s;
        }
        f32x16 gc, ge0, ge1, ge2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { gc[r] = 0.f; ge0[r] = 0.f; ge1[r] = 0.f; ge2[r] = 0.f; }

#pragma unroll
        for (int i = 4; i >= 0; --i) {
            if (WGRAD && valid) stage_block(srow, ST::SGH(i), h, gh);
            // through fc_c[i]: d/d c += Wc_i^T gh   (own-grid channels = in-block 0 only)
            mfma_chain_T(gc, lds + L::P_WC(i), loT, h, gh);
            // through relu
            f32x16 gp;
#pragma unroll
            for (int r = 0; r < 16; ++r) gp[r] = (mask[i] >> r) & 1u ? gh[r] : 0.f;
            if (WGRAD && valid) stage_block(srow, ST::SGP(i), h, gp);
            if (i == 0) {
                if (NEED_E) {
                    mfma_chain_T(ge0, lds + L::P_WP(0), loT, h, gp);
                    mfma_chain_T(ge1, lds + L::P_WP(0) + 4 * ADFP_SG, loT, h, gp);
                    mfma_chain_T(ge2, lds + L::P_WP(0) + 8 * ADFP_SG, loT, h, gp);
                }
            } else {
                f32x16 gn;
#pragma unroll
                for (int r = 0; r < 16; ++r) gn[r] = 0.f;
                if (i == 3) {
                    if (NEED_E) {
                        mfma_chain_T(ge0, lds + L::P_WP(3), loT, h, gp);
                        mfma_chain_T(ge1, lds + L::P_WP(3) + 4 * ADFP_SG, loT, h, gp);
                        mfma_chain_T(ge2, lds + L::P_WP(3) + 8 * ADFP_SG, loT, h, gp);
                    }
                    mfma_chain_T(gn, lds + L::P_WP(3) + L::chain_floats(L::KSE), loT, h, gp);
                } else mfma_chain_T(gn, lds + L::P_WP(i), loT, h, gp);
                gh = gn;
            }
        }
        if constexpr (WGRAD) if (valid) {       // d/d (p @ B) = d/d e * cos(p @ B)
            f32x16 t;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = ge0[r] * ce[r];
            stage_block(srow, ST::SGA, h, t);
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = ge1[r] * ce[16 + r];
            stage_block(srow, ST::SGA + 32, h, t);
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = ge2[r] * ce[32 + r];
            stage_block(srow, ST::SGA + 64, h, t);
        }

        if constexpr (PGRAD) {
            // through the Fourier features: d/dp_k = sum_j B[k][j] cos(p @ B)_j d/d e_j
            float gp[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < L::KSE; ++s) {
                const f32x4 bm = *(const f32x4*)(lds + L::P_BM + (unit_of(s, 0) + 4 * h) * 4);
                const float ga = (s < 16 ? ge0[s] : (s < 32 ? ge1[s - 16] : ge2[s - 32])) * ce[s];
                gp[0] = fmaf(ga, bm.x, gp[0]); gp[1] = fmaf(ga, bm.y, gp[1]); gp[2] = fmaf(ga, bm.z, gp[2]);
            }
            // through the trilinear feature lookup of the OWN grid (the high decoder's low-grid features
            // are under no_grad in the reference, decoder.py:182-187)
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
            gp[0] = fmaf(gx, dcx, gp[0]); gp[1] = fmaf(gy, dcy, gp[1]); gp[2] = fmaf(gz, dcz, gp[2]);
#pragma unroll
            for (int k = 0; k < 3; ++k) gp[k] += __shfl_xor(gp[k], 32);
            if (valid && h == 0) {
                a.g_pts[3ll * q + 0] += gp[0]; a.g_pts[3ll * q + 1] += gp[1]; a.g_pts[3ll * q + 2] += gp[2];
            }
        }
        // ---------------- scatter d/d c into the channels-last grid gradient ----------------
        if (a.g_grid) scatter_tile<CACHE>(a.g_grid, a.g0, pn, valid, gc, lane, sm);
    }
    if (a.g_grid) scatter_flush<CACHE>(a.g_grid, lane, sm);
}

// ------------------------------------------------------------------------------------------
// attention (mlp_tsdf) backward on the in-band list
// ------------------------------------------------------------------------------------------
struct AttBwdArgs {
    const float* packed; const int* list; const int* count_ptr;
    const float* att_occ; const float* att_u;
    const float* g_weight;     // [P] cotangent of the attention weight output (or NULL)
    const int* skip;           // device flag: non-zero = zero gradients for this call (see k_composite_bwd)
    float* g_raw;              // [P,4]: .w read as cotangent of the fused occupancy, then overwritten
                               //        with d/d(high+low) for the LOW backward
    float* att_g;              // per list entry: d/d(high+low) for the HIGH backward
    float* stage;
    int chunk_lo, chunk_hi;
    // PGRAD: the fused occupancy depends on the sample position through u = inv_tsdf(tsdf(p)) (decoder.py:241-248)
    PtsDev P; NormDev nt; TsdfDev t; float* g_pts;
    const float* gmax;         // non-NULL: the staged GRADIENT blocks are multiplied by grad_scale(gmax) (for k_outer_h)
    unsigned* dbg_masks;       // debug export of the recomputed ReLU decisions + softmax weights (adfp_train_state.dbg_masks_att) or NULL
};

template <bool WGRAD, bool PGRAD>
__global__ __launch_bounds__(256) void k_attention_bwd(AttBwdArgs a) {
    using A = AttLayout;
    using ST = AttStage;
    __shared__ __attribute__((aligned(16))) float lds[A::P_TOTAL];
    for (int i = threadIdx.x; i < A::P_TOTAL / 4; i += 256) ((f32x4*)lds)[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
    const int lane_off = h * ADFP_RG + p * 4;
    const int loT = lane_off_T(p);
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int cnt = *a.count_ptr;
    const float gS = WGRAD ? grad_scale(a.gmax) : 1.f;
    const int hi = a.chunk_hi < cnt ? a.chunk_hi : cnt;
    const int count = hi - a.chunk_lo;
    const int ntiles = count > 0 ? (count + 31) >> 5 : 0;
    for (int tile = wave; tile < ntiles; tile += nwaves) {
        const int loc = tile * 32 + p;
        const bool valid = loc < count;
        const int idx = a.chunk_lo + (valid ? loc : 0);
        const int q = a.list[idx];
        float* srow = WGRAD ? a.stage + (long long)loc * ST::NCOLS : nullptr;
        const float occ = a.att_occ[idx], u = a.att_u[idx];
        // ---- forward recompute with ReLU masks
        float h0[32];
        unsigned m0 = 0;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const f32x4 t = *(const f32x4*)(lds + A::P_A0 + (unit_of(s, 0) + 4 * h) * 4);
            h0[s] = fmaxf(fmaf(u, t.y, fmaf(occ, t.x, t.z)), 0.f);
            m0 |= h0[s] > 0.f ? (1u << s) : 0u;
        }
        if (WGRAD && valid) {
            stage_head(srow, ST::AX, h, f32x4{occ, u, 1.f, 0.f});
            stage_block(srow, ST::AH0, h, h0, 0); stage_block(srow, ST::AH0 + 32, h, h0, 16);
        }
        float h1[64]; unsigned m1[4];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            f32x16 acc;
            bias_init(acc, lds + A::P_B1 + 32 * ob, h);
            mfma_chain<32>(acc, lds + A::P_W1 + ob * A::BLK1, lane_off, h0);
            m1[ob] = pos_mask(acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) h1[16 * ob + r] = fmaxf(acc[r], 0.f);
            if (WGRAD && valid) stage_block(srow, ST::AH1 + 32 * ob, h, h1, 16 * ob);
        }
        float h2[64]; unsigned m2[4];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            f32x16 acc;
            bias_init(acc, lds + A::P_B2 + 32 * ob, h);
            mfma_chain<64>(acc, lds + A::P_W2 + ob * A::BLK2, lane_off, h1);
            m2[ob] = pos_mask(acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) h2[16 * ob + r] = fmaxf(acc[r], 0.f);
            if (WGRAD && valid) stage_block(srow, ST::AH2 + 32 * ob, h, h2, 16 * ob);
        }
        float l0 = 0.f, l1 = 0.f; unsigned m3[2];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            f32x16 acc;
            bias_init(acc, lds + A::P_B3 + 32 * ob, h);
            mfma_chain<64>(acc, lds + A::P_W3 + ob * A::BLK2, lane_off, h2);
            m3[ob] = pos_mask(acc);
            const float* w0 = lds + A::P_WO + (h * 2 + 0) * 32 + 16 * ob;
            const float* w1 = lds + A::P_WO + (h * 2 + 1) * 32 + 16 * ob;
            f32x16 h3;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                h3[r] = fmaxf(acc[r], 0.f);
                l0 = fmaf(h3[r], w0[r], l0);
                l1 = fmaf(h3[r], w1[r], l1);
            }
            if (WGRAD && valid) stage_block(srow, ST::AH3 + 32 * ob, h, h3);
        }
        l0 += __shfl_xor(l0, 32); l1 += __shfl_xor(l1, 32);
        l0 += lds[A::P_BO]; l1 += lds[A::P_BO + 1];
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
        const float den = e0 + e1;
        const float a0 = e0 / den, a1 = e1 / den;
        if (a.dbg_masks && valid) {     // k_attention_h<TRAIN>'s layout: value v of a layer at bit 31 - v, counted over the layer's words
            unsigned* mrow = a.dbg_masks + ((long long)idx * 2 + h) * 7;
            mrow[0] = __brev(m0);
            mrow[1] = (__brev(m1[0]) & 0xFFFF0000u) | (__brev(m1[1]) >> 16); mrow[2] = (__brev(m1[2]) & 0xFFFF0000u) | (__brev(m1[3]) >> 16);
            mrow[3] = (__brev(m2[0]) & 0xFFFF0000u) | (__brev(m2[1]) >> 16); mrow[4] = (__brev(m2[2]) & 0xFFFF0000u) | (__brev(m2[3]) >> 16);
            mrow[5] = (__brev(m3[0]) & 0xFFFF0000u) | (__brev(m3[1]) >> 16);
            mrow[6] = __float_as_uint(h ? a1 : a0);
        }
        // ---- backward: out = a0 occ + a1 u, w = a1
        const float g_out = valid ? a.g_raw[4ll * q + 3] : 0.f;
        const float g_w = (valid && a.g_weight && !(a.skip && *a.skip)) ? a.g_weight[q] : 0.f;
        const float ga0 = g_out * occ, ga1 = g_out * u + g_w;
        const float dot = a0 * ga0 + a1 * ga1;
        const float gl0 = a0 * (ga0 - dot), gl1 = a1 * (ga1 - dot);
        if (WGRAD && valid) stage_head(srow, ST::AGL, h, f32x4{gl0 * gS, gl1 * gS, 0.f, 0.f});
        float gp3[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float g = fmaf(lds[A::P_WO + (h * 2 + 0) * 32 + j], gl0, lds[A::P_WO + (h * 2 + 1) * 32 + j] * gl1);
            gp3[j] = (m3[j >> 4] >> (j & 15)) & 1u ? g : 0.f;
        }
        if (WGRAD && valid) { stage_block_mul(srow, ST::AG3, h, gp3, 0, gS); stage_block_mul(srow, ST::AG3 + 32, h, gp3, 16, gS); }
        float gp2[64];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
                mfma_chain_T(acc, lds + A::P_W3 + ob * A::BLK2 + ib * 4 * ADFP_SG, loT, h, gp3, 16 * ob);
#pragma unroll
            for (int r = 0; r < 16; ++r) gp2[16 * ib + r] = (m2[ib] >> r) & 1u ? acc[r] : 0.f;
            if (WGRAD && valid) stage_block_mul(srow, ST::AG2 + 32 * ib, h, gp2, 16 * ib, gS);
        }
        float gp1[64];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
                mfma_chain_T(acc, lds + A::P_W2 + ob * A::BLK2 + ib * 4 * ADFP_SG, loT, h, gp2, 16 * ob);
#pragma unroll
            for (int r = 0; r < 16; ++r) gp1[16 * ib + r] = (m1[ib] >> r) & 1u ? acc[r] : 0.f;
            if (WGRAD && valid) stage_block_mul(srow, ST::AG1 + 32 * ib, h, gp1, 16 * ib, gS);
        }
        float gx = 0.f, gxu = 0.f;
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
                mfma_chain_T(acc, lds + A::P_W1 + ob * A::BLK1 + ib * 4 * ADFP_SG, loT, h, gp1, 16 * ob);
            f32x16 g0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                g0[r] = (m0 >> (16 * ib + r)) & 1u ? acc[r] : 0.f;
                gx = fmaf(lds[A::P_A0 + (32 * ib + kmapH(r, h)) * 4], g0[r], gx);     // d/d occ_in through layer 0
                if (PGRAD) gxu = fmaf(lds[A::P_A0 + (32 * ib + kmapH(r, h)) * 4 + 1], g0[r], gxu);   // d/d u
            }
            if (WGRAD && valid) stage_block_mul(srow, ST::AG0 + 32 * ib, h, g0, 0, gS);
        }
        gx += __shfl_xor(gx, 32);
        const float g_in = a0 * g_out + gx;
        if (valid && h == 0) { a.att_g[idx] = g_in; a.g_raw[4ll * q + 3] = g_in; }
        if constexpr (PGRAD) {
            gxu += __shfl_xor(gxu, 32);
            const float g_u = a1 * g_out + gxu;
            if (valid && h == 0) {
                double pt[3]; float pn[3], dn[3], gt[3];
                load_point(a.P, q, pt);
                normalize3(a.nt, pt, pn);
#pragma unroll
                for (int k = 0; k < 3; ++k) dn[k] = (float)(2.0 * a.nt.inv[k]);
                const float tv = trilerp_scalar_grad(a.t, pn, dn, gt);
                // u = clamp(-0.1 log(1/(s + 1e-8) - 1 + 1e-7), +-100), s = clamp(1 - (t + 1)/2, 0, 1)
                const float sr = 1.f - (tv + 1.f) / 2.f;
                const float sc = fminf(fmaxf(sr, 0.f), 1.f);
                const float se = sc + 1e-8f;
                const float vv = (1.f / se) - 1.f + 1e-7f;
                const float ur = -0.1f * logf(vv);
                float du_dt = (sr > 0.f && sr < 1.f && ur > -100.f && ur < 100.f) ? -0.05f / (vv * se * se) : 0.f;
                const float g_t = g_u * du_dt;
                a.g_pts[3ll * q + 0] += g_t * gt[0]; a.g_pts[3ll * q + 1] += g_t * gt[1]; a.g_pts[3ll * q + 2] += g_t * gt[2];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// weight gradients: one 32x32 outer-product block per (job, point-chunk) wave
// ------------------------------------------------------------------------------------------
struct OuterJob { int colA, colB, dst, rs, cs, nr, j0, nc; };
#define OUTER_MAX_JOBS 56
struct OuterArgs {
    const float* stage; int ncols;
    const int* count_ptr; int chunk_lo, chunk_hi;   // rows = min(chunk_hi, *count_ptr) - chunk_lo
    float* flat;                                    // flat gradient (state_dict order)
    float* partial; int part_stride;                // k_outer_lds: one private copy of the flat gradient per workgroup
    int njobs; int rows_per_wave;
    OuterJob jobs[OUTER_MAX_JOBS];
};
__global__ __launch_bounds__(64) void k_outer(OuterArgs a) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    int hi = a.chunk_hi;
    if (a.count_ptr) { const int cnt = *a.count_ptr; hi = hi < cnt ? hi : cnt; }
    const int rows = hi - a.chunk_lo;
    const int m0 = blockIdx.x * a.rows_per_wave;
    if (m0 >= rows) return;
    const int m1 = (m0 + a.rows_per_wave < rows) ? m0 + a.rows_per_wave : rows;
    const OuterJob jb = a.jobs[blockIdx.y];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int m = m0; m < m1; m += 32) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int pt = m + 2 * s + h;
            float va = 0.f, vb = 0.f;
            if (pt < m1) {
                const float* row = a.stage + (long long)pt * a.ncols;
                va = row[jb.colA + i];
                vb = row[jb.colB + i];
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va, vb, acc, 0, 0, 0);
        }
    }
    const int j = i - jb.j0;
    if (j >= 0 && j < jb.nc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = kmapH(r, h);
            if (row < jb.nr && acc[r] != 0.f) atomicAdd(a.flat + jb.dst + row * jb.rs + j * jb.cs, acc[r]);
        }
    }
}

// The same job table, staged through LDS.  k_outer reads every staging column block once per job that
// uses it (2 x 128 B per point per job, ~7.7 KB per point for a decoder whose row is 3 KB) and runs at the
// rate L2 delivers those 128-B pieces.  Here a 512-thread workgroup streams its rows ONCE, 16 at a time,
// coalesced into a double-buffered LDS tile of whole staging rows, and its 8 waves split the jobs (up to 7
// accumulator blocks per wave); every MFMA operand is a conflict-free ds_read_b32.  One barrier per 16 rows.
// Two jobs may write the same gradient element only through DIFFERENT (row, column) sub-blocks, so within a
// workgroup the slot updates never collide.
#define OUTER_NW 8
#define OUTER_JW (OUTER_MAX_JOBS / OUTER_NW)
#define OUTER_RT 16
#define OUTER_MAXCOLS 832
__global__ __launch_bounds__(512) void k_outer_lds(OuterArgs a) {
    __shared__ __attribute__((aligned(16))) float sm[2 * OUTER_RT * OUTER_MAXCOLS];
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    int hi = a.chunk_hi;
    if (a.count_ptr) { const int cnt = *a.count_ptr; hi = hi < cnt ? hi : cnt; }
    const int rows = hi - a.chunk_lo;
    const int m0 = blockIdx.x * a.rows_per_wave;          // rows_per_wave = rows per WORKGROUP here
    if (m0 >= rows) return;
    const int m1 = (m0 + a.rows_per_wave < rows) ? m0 + a.rows_per_wave : rows;
    const int nc = a.ncols, n4 = OUTER_RT * nc / 4;       // float4 pieces of one 16-row tile
    constexpr int LD = (OUTER_RT * OUTER_MAXCOLS / 4 + 511) / 512;
    f32x16 acc[OUTER_JW];
#pragma unroll
    for (int j = 0; j < OUTER_JW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    int ca[OUTER_JW], cb[OUTER_JW];                        // this wave's jobs: byte-free LDS column offsets
#pragma unroll
    for (int j = 0; j < OUTER_JW; ++j) {
        const int job = wv + OUTER_NW * j;
        ca[j] = job < a.njobs ? a.jobs[job].colA + i : -1;
        cb[j] = job < a.njobs ? a.jobs[job].colB + i : -1;
    }
    f32x4 ld[LD];
    auto fetch = [&](int row0) {                           // rows row0 .. row0+15 -> registers (zeros past m1)
        const f32x4* src = (const f32x4*)(a.stage + (long long)row0 * nc);
        const int lim = (m1 - row0) * (nc / 4);            // float4 pieces that belong to real rows
#pragma unroll
        for (int k = 0; k < LD; ++k) {
            const int e = threadIdx.x + 512 * k;
            ld[k] = (e < n4 && e < lim) ? src[e] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stash = [&](int buf) {
        f32x4* dst = (f32x4*)(sm + buf * OUTER_RT * OUTER_MAXCOLS);
#pragma unroll
        for (int k = 0; k < LD; ++k) { const int e = threadIdx.x + 512 * k; if (e < n4) dst[e] = ld[k]; }
    };
    fetch(m0);
    stash(0);
    int buf = 0;
    for (int m = m0; m < m1; m += OUTER_RT, buf ^= 1) {
        const bool more = m + OUTER_RT < m1;
        if (more) fetch(m + OUTER_RT);
        __syncthreads();                                   // tile `buf` complete; everyone is done with `buf ^ 1`
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
    // The workgroup owns slot blockIdx.x of `partial` (zeroed before the network's first chunk): plain
    // read-modify-write, no atomics -- 250+ workgroups adding 64-134 KB each into ONE copy of the gradient ran
    // at a fraction of the atomic rate (all adders on the same few rows), and the sum is now reproducible.
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

// flat[e] += sum over the workgroup slots of partial[slot][e]
__global__ __launch_bounds__(256) void k_reduce_partials(const float* __restrict__ partial, int nslots, int stride, int n,
                                                         float* __restrict__ flat) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 4 <= nslots; k += 4) {
        s0 += partial[(long long)k * stride + e]; s1 += partial[(long long)(k + 1) * stride + e];
        s2 += partial[(long long)(k + 2) * stride + e]; s3 += partial[(long long)(k + 3) * stride + e];
    }
    for (; k < nslots; ++k) s0 += partial[(long long)k * stride + e];
    flat[e] += (s0 + s1) + (s2 + s3);
}

// d/d rays_o = sum_s d/dp_s,  d/d rays_d = sum_s z_s d/dp_s   (p = o + d z, Renderer.py:223)
// extra / n_extra: further [P,3] buffers (the decoders' own, k_decode_bwd_h_pgrad3) added to g_pts in order, in float like the
// accumulation in place they replace
struct RaysGradExtra { const float* p[3]; int n; };
__global__ __launch_bounds__(256) void k_rays_grad(const float* __restrict__ g_pts, const double* __restrict__ z, int n_rays, int S,
                                                   float* __restrict__ g_o, float* __restrict__ g_d, RaysGradExtra extra) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    double so[3] = {0, 0, 0}, sd[3] = {0, 0, 0};
    for (int s = lane; s < S; s += 64) {
        const long long q = (long long)ray * S + s;
        const double zz = z[q];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float gf = g_pts[3 * q + k];
            for (int e = 0; e < extra.n; ++e) gf += extra.p[e][3 * q + k];
            const double g = (double)gf; so[k] += g; sd[k] += g * zz;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) { so[k] += __shfl_xor(so[k], o); sd[k] += __shfl_xor(sd[k], o); }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { if (g_o) g_o[3 * ray + k] = (float)so[k]; if (g_d) g_d[3 * ray + k] = (float)sd[k]; }
    }
}
