// adfp_kernels.hip -- gfx950 kernels + the C ABI of libadfp.so (see include/adfp.h).
//
// Forward pipeline of Renderer.render_batch_ray (reference src/utils/Renderer.py:110-255):
//
//   k_depth_max   batch-global max(gt_depth)                      Renderer.py:159, :195
//   k_sample      z_vals[N,S] f64: uniform + surface, rank-merged  Renderer.py:163-221
//   k_tsdf        TSDF trilerp, band mask, compaction of in-band   decoder.py:295-303, :329
//                 points (the HBM-streaming stage)
//   k_decode<LOW>   low decoder on every point (MFMA f32)           decoder.py:177-203
//   k_decode<COLOR> colour decoder on every point (stage color)
//   k_decode<HIGH>  high decoder on the in-band list only -- its output is dead
//                   everywhere else (decoder.py:333: unmasked points keep `low`)
//   k_attention   mlp_tsdf on the in-band list (MFMA f32)          decoder.py:240-258
//   k_composite   sigmoid(10 occ), transmittance scan, sums         common.py:234-251
//
// Each MFMA kernel keeps ONE network's weights resident in LDS (64-134 KB) in MFMA operand
// order and chains layers accumulator -> next B operand without leaving registers.
#include "adfp_device.h"
#include <math.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#define ADFP_CHECK_LAUNCH()                         \
    do {                                            \
        hipError_t e_ = hipGetLastError();          \
        if (e_ != hipSuccess) return (int)e_;       \
    } while (0)

// Zero-fill as a KERNEL, not hipMemsetAsync: every device write of a call is then an ordinary kernel node when the
// caller captures the call into a HIP graph (torch.cuda.graph); with memset nodes in the captured sequence, replays
// interleaved with other allocations faulted on ROCm 7.2 (tools/graph_capture.py).  bytes must be a multiple of 4.
__global__ __launch_bounds__(256) void k_zero(unsigned* __restrict__ p, long long n_words) {
    const long long n4 = n_words >> 2;
    const long long stride = (long long)gridDim.x * 256;
    if ((((unsigned long long)p) & 15ull) == 0) {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) ((uint4*)p)[i] = uint4{0u, 0u, 0u, 0u};
        for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n_words; i += stride) p[i] = 0u;
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_words; i += stride) p[i] = 0u;
    }
}
// several buffers in ONE launch
#define ZERO_MULTI 8
struct ZeroJobs { unsigned* p[ZERO_MULTI]; long long words[ZERO_MULTI]; unsigned first_block[ZERO_MULTI + 1]; int n; };
__device__ inline void zero_multi_block(const ZeroJobs& z, unsigned blk) {
    int j = 0;
    while (j + 1 < z.n && blk >= z.first_block[j + 1]) ++j;
    unsigned* p = z.p[j];
    const long long n_words = z.words[j], n4 = n_words >> 2;
    const long long b = blk - z.first_block[j], stride = (long long)(z.first_block[j + 1] - z.first_block[j]) * 256;
    if ((((unsigned long long)p) & 15ull) == 0) {
        for (long long i = b * 256 + threadIdx.x; i < n4; i += stride) ((uint4*)p)[i] = uint4{0u, 0u, 0u, 0u};
        for (long long i = (n4 << 2) + b * 256 + threadIdx.x; i < n_words; i += stride) p[i] = 0u;
    } else {
        for (long long i = b * 256 + threadIdx.x; i < n_words; i += stride) p[i] = 0u;
    }
}
__global__ __launch_bounds__(256) void k_zero_multi(ZeroJobs z) { zero_multi_block(z, blockIdx.x); }
struct ZeroBatch {
    ZeroJobs z; unsigned blocks;
    ZeroBatch() : blocks(0) { z.n = 0; z.first_block[0] = 0; }
    hipError_t flush(hipStream_t st) {
        if (z.n == 0) return hipSuccess;
        hipLaunchKernelGGL(k_zero_multi, dim3(blocks), dim3(256), 0, st, z);
        z.n = 0; blocks = 0; z.first_block[0] = 0;
        return hipGetLastError();
    }
    hipError_t add(void* p, size_t bytes, hipStream_t st) {
        if (!p || bytes == 0) return hipSuccess;
        if (z.n == ZERO_MULTI) { hipError_t e = flush(st); if (e != hipSuccess) return e; }
        const long long words = (long long)(bytes / 4);
        long long b = (words / 4 + 255) / 256;
        b = b < 1 ? 1 : (b > 1024 ? 1024 : b);
        z.p[z.n] = (unsigned*)p; z.words[z.n] = words; blocks += (unsigned)b; z.first_block[++z.n] = blocks;
        return hipSuccess;
    }
};
static hipError_t zero_async(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    const long long words = (long long)(bytes / 4);
    long long blocks = (words / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_zero, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned*)p, words);
    return hipGetLastError();
}


// =====================================================================================
// layout conversion
// =====================================================================================
// [C=32][V] -> [V][32], through a padded LDS tile so both sides are coalesced.
__global__ __launch_bounds__(256) void k_relayout_cm_to_cl(const float* __restrict__ src, float* __restrict__ dst, long long V) {
    __shared__ float tile[32][65];
    const long long v0 = (long long)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
#pragma unroll
    for (int c = ty; c < 32; c += 4) {
        const long long v = v0 + tx;
        tile[c][tx] = v < V ? src[(long long)c * V + v] : 0.f;
    }
    __syncthreads();
    const int cx = threadIdx.x & 31, vy = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int j = vy; j < 64; j += 8) {
        const long long v = v0 + j;
        if (v < V) dst[v * 32 + cx] = tile[cx][j];
    }
}
// the same two conversions as block functions of a multi-job launch (k_relayout_multi, k_forward_head): `tile` = 32 x 65 floats of LDS
struct RelayoutJobs { int n; unsigned first[ADFP_RELAYOUT_MAX_JOBS + 1]; const float* src[ADFP_RELAYOUT_MAX_JOBS]; float* dst[ADFP_RELAYOUT_MAX_JOBS];
                      long long V[ADFP_RELAYOUT_MAX_JOBS]; };
template <bool BACK>
__device__ inline void relayout_multi_block(const RelayoutJobs& j, unsigned blk, float (*tile)[65]) {
    int k = 0;
    while (k + 1 < j.n && blk >= j.first[k + 1]) ++k;                     // block-uniform
    const float* __restrict__ src = j.src[k];
    float* __restrict__ dst = j.dst[k];
    const long long V = j.V[k], v0 = (long long)(blk - j.first[k]) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
    const int cx = threadIdx.x & 31, vy = threadIdx.x >> 5;   // 32 x 8
    if (!BACK) {
#pragma unroll
        for (int c = ty; c < 32; c += 4) { const long long v = v0 + tx; tile[c][tx] = v < V ? src[(long long)c * V + v] : 0.f; }
        __syncthreads();
#pragma unroll
        for (int jv = vy; jv < 64; jv += 8) { const long long v = v0 + jv; if (v < V) dst[v * 32 + cx] = tile[cx][jv]; }
    } else {
#pragma unroll
        for (int jv = vy; jv < 64; jv += 8) { const long long v = v0 + jv; tile[cx][jv] = v < V ? src[v * 32 + cx] : 0.f; }
        __syncthreads();
#pragma unroll
        for (int c = ty; c < 32; c += 4) { const long long v = v0 + tx; if (v < V) dst[(long long)c * V + v] = tile[c][tx]; }
    }
}
template <bool BACK>
__global__ __launch_bounds__(256) void k_relayout_multi(RelayoutJobs j) {
    __shared__ float tile[32][65];
    relayout_multi_block<BACK>(j, blockIdx.x, tile);
}
__global__ __launch_bounds__(256) void k_relayout_cl_to_cm(const float* __restrict__ src, float* __restrict__ dst, long long V) {
    __shared__ float tile[32][65];
    const long long v0 = (long long)blockIdx.x * 64;
    const int cx = threadIdx.x & 31, vy = threadIdx.x >> 5;
#pragma unroll
    for (int j = vy; j < 64; j += 8) {
        const long long v = v0 + j;
        tile[cx][j] = v < V ? src[v * 32 + cx] : 0.f;
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int c = ty; c < 32; c += 4) {
        const long long v = v0 + tx;
        if (v < V) dst[(long long)c * V + v] = tile[c][tx];
    }
}

// =====================================================================================
// weight packing: flat state_dict order -> MFMA operand order
// =====================================================================================
// position inside a padded chain image -> (k-step s, lane-half h, out row); row < 0 = padding
__device__ __forceinline__ void chain_pos(int u, int& s, int& h, int& row) {
    const int s4 = u / ADFP_SG, rem = u % ADFP_SG;
    h = rem / ADFP_RG;
    const int r2 = rem % ADFP_RG;
    row = r2 < 128 ? (r2 >> 2) : -1;
    s = s4 * 4 + (r2 & 3);
}

template <int CDIM, int NOUT>
__device__ int dec_src_index(int t) {
    using L = DecLayout<CDIM, NOUT>;
    if (t < 384) {
        const int j = t >> 2, c = t & 3;
        return (j < 93 && c < 3) ? L::F_EB + c * 93 + j : -1;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        if (t < L::P_BP(i)) {                       // pts_linears.i weight chain
            int s, h, row; chain_pos(t - L::P_WP(i), s, h, row);
            if (row < 0) return -1;
            int col = unit_of(s, h);                 // layer 0: feature; 1,2,4: hidden; 3: [feature(96 pad), hidden]
            if (i == 0) { if (col >= 93) return -1; }
            else if (i == 3) { if (col < 96) { if (col >= 93) return -1; } else col = 93 + (col - 96); }
            return L::F_PL(i) + row * L::in_dim(i) + col;
        }
        if (t < L::P_WC(i)) return L::F_PL(i) + 32 * L::in_dim(i) + (t - L::P_BP(i));
        if (t < L::P_BC(i)) {                       // fc_c.i weight chain
            int s, h, row; chain_pos(t - L::P_WC(i), s, h, row);
            if (row < 0) return -1;
            return L::F_FC(i) + row * CDIM + unit_of(s, h);
        }
        if (t < L::P_BC(i) + 32) return L::F_FC(i) + 32 * CDIM + (t - L::P_BC(i));
    }
    if (t < L::P_BO) {
        const int u = t - L::P_WO;
        const int h = u / (NOUT * 16), o = (u >> 4) % NOUT, r = u & 15;
        return L::F_OW + o * 32 + kmapH(r, h);
    }
    const int o = t - L::P_BO;
    return o < NOUT ? L::F_OB + o : -1;
}

template <int CDIM, int NOUT>
__global__ void k_pack_decoder(const float* __restrict__ flat, float* __restrict__ packed) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= DecLayout<CDIM, NOUT>::P_TOTAL) return;
    const int s = dec_src_index<CDIM, NOUT>(t);
    packed[t] = s < 0 ? 0.f : flat[s];
}
// gradient of the packed image -> gradient of the flat parameters (every flat element has
// exactly one packed position)
template <int CDIM, int NOUT>
__global__ void k_unpack_decoder_grad(const float* __restrict__ packed, float* __restrict__ flat) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= DecLayout<CDIM, NOUT>::P_TOTAL) return;
    const int s = dec_src_index<CDIM, NOUT>(t);
    if (s >= 0) flat[s] = packed[t];
}

__device__ int att_src_index(int t) {
    using A = AttLayout;
    if (t < A::P_W1) {
        const int k = t >> 2, c = t & 3;
        return c < 2 ? A::F_W0 + k * 2 + c : (c == 2 ? A::F_B0 + k : -1);
    }
    if (t < A::P_B1) {                              // 64 -> 128
        const int u = t - A::P_W1, blk = u / A::BLK1;
        int s, h, row; chain_pos(u % A::BLK1, s, h, row);
        return row < 0 ? -1 : A::F_W1 + (blk * 32 + row) * 64 + unit_of(s, h);
    }
    if (t < A::P_W2) return A::F_B1 + (t - A::P_B1);
    if (t < A::P_B2) {                              // 128 -> 128
        const int u = t - A::P_W2, blk = u / A::BLK2;
        int s, h, row; chain_pos(u % A::BLK2, s, h, row);
        return row < 0 ? -1 : A::F_W2 + (blk * 32 + row) * 128 + unit_of(s, h);
    }
    if (t < A::P_W3) return A::F_B2 + (t - A::P_B2);
    if (t < A::P_B3) {                              // 128 -> 64
        const int u = t - A::P_W3, blk = u / A::BLK2;
        int s, h, row; chain_pos(u % A::BLK2, s, h, row);
        return row < 0 ? -1 : A::F_W3 + (blk * 32 + row) * 128 + unit_of(s, h);
    }
    if (t < A::P_WO) return A::F_B3 + (t - A::P_B3);
    if (t < A::P_BO) {                              // [h][o][32]: entry j <-> unit_of(j, h)
        const int u = t - A::P_WO, h = u >> 6, o = (u >> 5) & 1, j = u & 31;
        return A::F_WO + o * 64 + unit_of(j, h);
    }
    const int o = t - A::P_BO;
    return o < 2 ? A::F_BO + o : -1;
}
__global__ void k_pack_attention(const float* __restrict__ flat, float* __restrict__ packed) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= AttLayout::P_TOTAL) return;
    const int s = att_src_index(t);
    packed[t] = s < 0 ? 0.f : flat[s];
}

// =====================================================================================
// a1: get_rays (common.py:254-272)
// =====================================================================================
__global__ void k_get_rays(int H, int W, float fx, float fy, float cx, float cy, const float* __restrict__ c2w,
                           float* __restrict__ ro, float* __restrict__ rd) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const int j = idx / W, i = idx - j * W;
    // torch.linspace(0, W-1, W) is exactly the integers; dirs = ((i-cx)/fx, -(j-cy)/fy, -1)
    const float dx = ((float)i - cx) / fx, dy = -((float)j - cy) / fy, dz = -1.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // torch.sum(dirs * c2w[:3,:3], -1): products then left-to-right adds
        const float s = __fadd_rn(__fadd_rn(__fmul_rn(dx, c2w[4 * k + 0]), __fmul_rn(dy, c2w[4 * k + 1])),
                                  __fmul_rn(dz, c2w[4 * k + 2]));
        rd[3 * idx + k] = s;
        ro[3 * idx + k] = c2w[4 * k + 3];
    }
}

// =====================================================================================
// a2: get_rays_from_uv (common.py:76-91): the rays through n given pixels, and the gradient w.r.t. the camera pose that
// the Tracker and the Mapper's bundle adjustment take through them (src/Tracker.py:97, src/Mapper.py:425)
// =====================================================================================
__global__ void k_rays_from_uv(const float* __restrict__ pi, const float* __restrict__ pj, int n, float fx, float fy, float cx, float cy,
                               const float* __restrict__ c2w, float* __restrict__ ro, float* __restrict__ rd) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float dx = (pi[idx] - cx) / fx, dy = -(pj[idx] - cy) / fy, dz = -1.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // torch.sum(dirs * c2w[:3,:3], -1): products then left-to-right adds
        rd[3 * idx + k] = __fadd_rn(__fadd_rn(__fmul_rn(dx, c2w[4 * k + 0]), __fmul_rn(dy, c2w[4 * k + 1])), __fmul_rn(dz, c2w[4 * k + 2]));
        ro[3 * idx + k] = c2w[4 * k + 3];
    }
}
// g_c2w[k][m] = sum_n g_d[n][k] * dirs[n][m] (m < 3), g_c2w[k][3] = sum_n g_o[n][k]; rows 3 of the 4x4 get zero.
// One workgroup: pixel batches are a few hundred to a few thousand rays.
__global__ __launch_bounds__(256) void k_rays_from_uv_bwd(const float* __restrict__ pi, const float* __restrict__ pj, int n, float fx, float fy,
                                                          float cx, float cy, const float* __restrict__ g_o, const float* __restrict__ g_d,
                                                          float* __restrict__ g_c2w) {
    __shared__ float s[4][12];
    float acc[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) acc[t] = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float dir[3] = {(pi[i] - cx) / fx, -(pj[i] - cy) / fy, -1.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float gd = g_d ? g_d[3 * i + k] : 0.f;
#pragma unroll
            for (int m = 0; m < 3; ++m) acc[4 * k + m] = fmaf(gd, dir[m], acc[4 * k + m]);
            acc[4 * k + 3] += g_o ? g_o[3 * i + k] : 0.f;
        }
    }
#pragma unroll
    for (int t = 0; t < 12; ++t) acc[t] = wave_sum(acc[t]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int t = 0; t < 12; ++t) s[threadIdx.x >> 6][t] = acc[t];
    __syncthreads();
    if (threadIdx.x < 16) g_c2w[threadIdx.x] = threadIdx.x < 12 ? (s[0][threadIdx.x] + s[1][threadIdx.x]) + (s[2][threadIdx.x] + s[3][threadIdx.x]) : 0.f;
}

// =====================================================================================
// a3: the Mapper's bounding-box pre-filter (Mapper.py:438-449): keep ray i iff
//   min_axis max_side((bound - o) / d) >= gt_depth          (f64, NaN compares false)
// Order-preserving compaction of the kept ray ids by ONE workgroup: a ballot prefix inside each
// wave, a 16-entry LDS scan across waves, a running base across 1024-ray rounds.  Mapping batches
// are a few thousand rays, so one workgroup is the whole job.
// =====================================================================================
__global__ __launch_bounds__(1024) void k_prefilter(const float* __restrict__ ro, const float* __restrict__ rd,
                                                    const float* __restrict__ depth, int n, const double* __restrict__ bnd,
                                                    int* __restrict__ out_index, int* __restrict__ out_count) {
    __shared__ int s_w[16];
    __shared__ int s_base;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double b[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) b[k] = bnd[k];
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        bool keep = false;
        if (i < n) {
            double t = INFINITY; bool nan = false;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double o = (double)ro[3 * i + k], d = (double)rd[3 * i + k];
                const double t0 = (b[2 * k] - o) / d, t1 = (b[2 * k + 1] - o) / d;
                nan |= (t0 != t0) | (t1 != t1);            // torch.max / torch.min propagate NaN
                const double tm = t0 > t1 ? t0 : t1;
                t = tm < t ? tm : t;
            }
            keep = !nan && (t >= (double)depth[i]);
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_w[wv] = __popcll(m);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int c = s_w[w]; before += w < wv ? c : 0; total += c; }
        const int base = s_base;
        if (keep) out_index[base + before + __popcll(m & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (threadIdx.x == 0) s_base = base + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *out_count = s_base;
}

// =====================================================================================
// a4: sampler
// =====================================================================================
#define SEGMAX_PARTS 16          // partial maxima per segment of rays: one block of the call's first launch each, folded by k_sample's lanes
#define SEGMAX_SEGS 48           // segments per call (render_img's ray batches)
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ __launch_bounds__(256) void k_depth_max(const float* __restrict__ d, int n, unsigned* __restrict__ out) {
    unsigned m = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned v = f2ord(d[i]);
        m = v > m ? v : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned v = __shfl_xor(m, o);
        m = v > m ? v : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// max(gt_depth) per segment of `seg` consecutive rays (render_img's ray batches): blockIdx.y = segment, blockIdx.x = slice of it;
// ordered-uint atomicMax into out[segment] (zeroed by the caller), like k_depth_max
__global__ __launch_bounds__(256) void k_depth_max_seg(const float* __restrict__ d, int n, int seg, unsigned* __restrict__ out) {
    const int lo = blockIdx.y * seg, hi = lo + seg < n ? lo + seg : n;
    unsigned m = 0;
    for (int i = lo + blockIdx.x * 256 + threadIdx.x; i < hi; i += gridDim.x * 256) {
        const unsigned v = f2ord(d[i]);
        m = v > m ? v : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned v = __shfl_xor(m, o);
        m = v > m ? v : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out + blockIdx.y, m);
}

struct SampleArgs {
    const float* ro; const float* rd; const float* depth; const float* t_rand;
    const float* dmax_f;        // device float(s) (caller-provided or k_depth_max_seg) or NULL
    int dmax_seg;               // > 0: dmax_f[(dmax_first + ray) / dmax_seg] (one maximum per segment of rays), 0: dmax_f[0]
    int dmax_first;             // the call's first ray in the segmented batch (a ray shard of a frame), else 0
    const unsigned* dmax_ord;   // ordered-uint reduction result or NULL
    const unsigned* dmax_parts; // or: SEGMAX_PARTS ordered-uint PARTIAL maxima per segment (k_forward_head's segment-maximum blocks), folded here
    double b[6];                // bound lo/hi per axis
    int n_rays, n_samples, n_surface, lindisp;
    float perturb;
    double* z;
    // side job of the launch (adfp_render_forward): blocks [nb_sample, gridDim.x) convert the feature grids the caller handed over
    // (adfp_render_args.relayout_jobs) to channels-last.  The sampler is a latency-bound launch that leaves most of the chip idle and
    // the decoders, the first readers of the grids, are two launches later; inside the call's FIRST launch the conversions (96 MB
    // moved per frame) stood in front of the sampler.
    int nb_sample; RelayoutJobs rl;
};

// torch.linspace(0, 1, n) for float32 (ATen RangeFactories: symmetric fill).  The upper half is
// end - step * (n - 1 - i); PyTorch's CPU build contracts it into one fma (the golden z_vals generated
// from the reference are reproduced to the last bit only with the fma), so it is an explicit fmaf here.
ADFP_DEV float linspace01(int i, int n) {
    if (n == 1) return 0.f;
    const float step = 1.0f / (float)(n - 1);
    return (i < n / 2) ? step * (float)i : fmaf(-step, (float)(n - i - 1), 1.0f);
}

struct RaySampler {
    float nearf; double neard; double far; float dep; double dmax; int ns, nf; bool has_depth, lindisp, perturb;
    const float* trand;
    // uniform sample k before perturbation (Renderer.py:203-208)
    ADFP_DEV double zu_raw(int k) const {
        const float t = linspace01(k, ns);
        if (!lindisp) {
            if (has_depth) return __dadd_rn((double)__fmul_rn(nearf, __fsub_rn(1.f, t)), __dmul_rn(far, (double)t));
            return __dadd_rn((double)__fmul_rn(0.01f, __fsub_rn(1.f, t)), __dmul_rn(far, (double)t));
        }
        if (has_depth) {
            const double a = (double)__fmul_rn(__fdiv_rn(1.f, nearf), __fsub_rn(1.f, t));
            return 1.0 / __dadd_rn(a, __dmul_rn(1.0 / far, (double)t));
        }
        return 1.0 / __dadd_rn((double)__fmul_rn(100.f, __fsub_rn(1.f, t)), __dmul_rn(1.0 / far, (double)t));
    }
    // with stratified jitter (Renderer.py:210-217)
    ADFP_DEV double zu(int k) const {
        const double z = zu_raw(k);
        if (!perturb) return z;
        const double lo = k == 0 ? z : .5 * (z + zu_raw(k - 1));
        const double up = k == ns - 1 ? z : .5 * (zu_raw(k + 1) + z);
        return lo + (up - lo) * (double)trand[k];
    }
    // surface sample j (Renderer.py:179-201)
    ADFP_DEV double zs(int j) const {
        const double t = (double)linspace01(j, nf);
        if (dep > 0.f) return __dadd_rn(__dmul_rn((double)__fmul_rn(0.95f, dep), 1.0 - t),
                                        __dmul_rn((double)__fmul_rn(1.05f, dep), t));
        return __dadd_rn(__dmul_rn(0.001, 1.0 - t), __dmul_rn(dmax, t));
    }
};

// One wave per ADFP_SAMPLE_RPW consecutive rays.  Lane e computes merged-list element e of the current ray (uniform e < ns, surface
// otherwise), stages it in the wave's LDS row, and finds its rank in the sorted union by counting (only the values matter:
// torch.sort, Renderer.py:220).
// Round 4: a wave used to take ONE ray -- origin / direction / depth loaded, one f64 division, 64 values, a binary search, one row
// written -- and the launch was bound by that chain's latency (160 us per 640 x 480 frame = 1 TB/s of z_vals written, 37 rounds of
// 32 waves per CU at ~4 us each).  Now the wave's first instruction loads the inputs of ALL its rays (16 lanes per ray: lane 16 r +
// sel takes axis sel >> 1, side sel & 1 of ray r), the one f64 division per lane serves four rays at once, and the rays are then
// worked off in turn out of registers (v_readlane with constant lane numbers) while the previous rows' stores drain.
#ifndef ADFP_SAMPLE_RPW
#define ADFP_SAMPLE_RPW 4
#endif
__global__ __launch_bounds__(256) void k_sample(SampleArgs a) {
    constexpr int RPW = ADFP_SAMPLE_RPW;
    static_assert(RPW == 1 || RPW == 2 || RPW == 4, "16 lanes per ray");
    if ((int)blockIdx.x >= a.nb_sample) {              // block-uniform: the launch's grid-conversion blocks
        __shared__ float tile[32][65];
        relayout_multi_block<false>(a.rl, blockIdx.x - (unsigned)a.nb_sample, tile);
        return;
    }
    __shared__ double sv[4][ADFP_MAX_SAMPLES];
    const int lane = threadIdx.x & 63;
    const int ray0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (ray0 >= a.n_rays) return;                      // wave-uniform; no block-wide barrier below
    double* v = sv[threadIdx.x >> 6];
    const bool has_depth = a.depth != nullptr;
    // ---- the inputs of the wave's rays, one load each, all in flight together
    const int myr = lane >> 4;                         // the ray (of the wave's RPW) this lane loads for
    const int lray = (myr < RPW && ray0 + myr < a.n_rays) ? ray0 + myr : ray0;
    float dep_l = 0.f, dmax_l = 0.f;
    double tmk_l;
    {
        const int sel = lane & 7, ax = sel < 6 ? sel >> 1 : 0, side = sel & 1;
        const double o = (double)a.ro[3 * lray + ax], d = (double)a.rd[3 * lray + ax];
        if (has_depth) {
            dep_l = a.depth[lray];
            const int sg = a.dmax_seg > 0 ? (a.dmax_first + lray) / a.dmax_seg : 0;
            if (a.dmax_parts) {                         // the 16 lanes of a ray fold the segment's 16 partial maxima (max is exact: any order)
                unsigned pm = a.dmax_parts[sg * SEGMAX_PARTS + (lane & 15)];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { const unsigned q = __shfl_xor(pm, o); pm = q > pm ? q : pm; }
                dmax_l = ord2f(pm);
            } else dmax_l = a.dmax_f ? a.dmax_f[sg] : ord2f(a.dmax_ord[sg]);
        }
        // far_bb = min_axis max_side (bound - o)/d + 0.01   (Renderer.py:151-156), f64: ONE division per lane -- lane (ray, axis, side)
        // (an f64 division is ~40 instructions; six of them in every lane were most of this kernel's VALU work).
        const double bk = ax == 0 ? (side ? a.b[1] : a.b[0]) : (ax == 1 ? (side ? a.b[3] : a.b[2]) : (side ? a.b[5] : a.b[4]));
        const double t = (bk - o) / d;
        const double tp = dpp_f64<0xB1>(t);              // lane ^ 1
        const double t0 = side ? tp : t, t1 = side ? t : tp;
        // torch.max / torch.min propagate NaN (0/0: a zero direction component with the origin on that bound
        // plane; inf/inf), and so does torch.clamp below -- the whole ray then samples NaN like the reference's
        tmk_l = (t0 != t0 || t1 != t1) ? (double)NAN : (t0 > t1 ? t0 : t1);
    }
    RaySampler rs;
    rs.ns = a.n_samples; rs.lindisp = a.lindisp != 0; rs.perturb = a.perturb > 0.f;
    rs.has_depth = has_depth;
    rs.nf = has_depth ? a.n_surface : 0;
    const int S = rs.ns + rs.nf;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int ray = ray0 + r;
        if (ray >= a.n_rays) break;                    // wave-uniform
        rs.trand = a.t_rand ? a.t_rand + (long long)ray * a.n_samples : nullptr;
        const float dmaxf = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dmax_l), 16 * r));
        rs.dmax = (double)dmaxf;
        rs.dep = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dep_l), 16 * r));
        rs.nearf = __fmul_rn(rs.dep, 0.01f);
        double far_bb = INFINITY;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double tm = readlane_f64(tmk_l, 16 * r + 2 * k);
            far_bb = (tm != tm || far_bb != far_bb) ? (double)NAN : (tm < far_bb ? tm : far_bb);
        }
        far_bb += 0.01;
        if (rs.has_depth) {
            const double hi = (double)__fmul_rn(dmaxf, 1.2f);
            double f = far_bb < 0.0 ? 0.0 : far_bb;       // clamp(far_bb, 0, max(gt_depth*1.2))
            rs.far = f > hi ? hi : f;
        } else rs.far = far_bb;
        // the row is the wave's own: LDS executes a wave's accesses in order, the fences keep the compiler from moving them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int e = lane; e < S; e += 64) v[e] = e < rs.ns ? rs.zu(e) : rs.zs(e - rs.ns);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double* zrow = a.z + (long long)ray * S;
        if (rs.nf == 0) {                                  // no sort in the reference either
            for (int e = lane; e < S; e += 64) zrow[e] = v[e];
            continue;
        }
        // Both lists are monotone (uniform samples run near -> far, surface samples 0.95 d -> 1.05 d), so
        // the rank of an element is its own index plus a binary-search count in the OTHER list (ties:
        // uniform first, like a stable sort of the concatenation).  The O(S^2) count is kept for the
        // degenerate cases: descending lists (far < near when the ray leaves the bound at once) and NaN samples
        // (NaN far plane; lindisp with a zero sensor depth gives inf * 0 in the last uniform sample), which
        // torch.sort orders after every number.
        bool any_nan = false;
        for (int e0 = 0; e0 < S; e0 += 64) { const int e = e0 + lane; any_nan |= __ballot(e < S && v[e] != v[e]) != 0ull; }
        const bool ascending = !any_nan && (v[0] <= v[rs.ns - 1]) && (v[rs.ns] <= v[S - 1]);
        if (ascending) {
            for (int e = lane; e < S; e += 64) {
                const double val = v[e];
                const bool uni = e < rs.ns;
                // uniform element: count surface elements <  val;  surface element: count uniform elements <= val
                int lo = uni ? rs.ns : 0, hi = uni ? S : rs.ns;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    const double o = v[mid];
                    const bool before = uni ? (o < val) : (o <= val);
                    if (before) lo = mid + 1; else hi = mid;
                }
                const int rank = uni ? e + (lo - rs.ns) : (e - rs.ns) + lo;
                zrow[rank] = val;
            }
            continue;
        }
        for (int e = lane; e < S; e += 64) {
            const double val = v[e];
            const bool vn = val != val;
            int rank = 0;
            for (int k = 0; k < S; ++k) {
                const double o = v[k];
                const bool on = o != o;
                const bool less = vn ? !on : (o < val);                  // every number sorts before a NaN
                const bool same = vn ? on : (o == val);
                rank += less || (same && k < e);
            }
            zrow[rank] = val;
        }
    }
}

// =====================================================================================
// TSDF stage (a10): trilerp + band mask + compaction.  One block = 2048 consecutive points;
// in-band points are staged in LDS and appended to the global list with ONE atomic per block.
// =====================================================================================
struct TsdfArgs {
    PtsDev P; NormDev nt; TsdfDev t; double b[6];
    unsigned char* flags; int* list; float* att_u; float* w; int* counter; float* tsdf_out;
    int rays_per_block;        // ray mode: RB, a power of two with RB * S <= TSDF_CHUNK (tsdf_rays_per_block)
};
#ifndef TSDF_CHUNK
#define TSDF_CHUNK 2048
#endif
// In ray mode a block owns RB consecutive rays (RB * S <= 2048 points) and walks them TRANSPOSED:
// consecutive lanes take the SAME sample index of consecutive rays.  Neighbouring pixels' samples of
// equal index are millimetres apart, so the 64 lanes of a load instruction fall into a few cache lines
// instead of 64 (consecutive samples of ONE ray are ~5 voxels apart) -- the stage is bound by the
// lines the address unit walks, not by bytes.  z_vals come in and flags / w go out through LDS so
// that global traffic stays coalesced.
#ifndef TSDF_UB
#define TSDF_UB 4
#endif
#ifndef TSDF_MINW
#define TSDF_MINW 4
#endif
__global__ __launch_bounds__(256, TSDF_MINW) void k_tsdf(TsdfArgs a) {
    constexpr int PER = TSDF_CHUNK / 256;            // points per thread
    constexpr int UB = TSDF_UB;                      // points whose loads are in flight together
    static_assert(PER % UB == 0, "TSDF_CHUNK must be a multiple of 256 * UB");
    __shared__ int s_q[TSDF_CHUNK];
    __shared__ float s_u[TSDF_CHUNK];
    __shared__ double s_z[TSDF_CHUNK + 64];
    // flags are written by lanes that hold the SAME sample of consecutive rays: with rows S bytes apart the 64 byte-stores of a
    // wave fell into 4 banks (S = 64: 16 dwords per row; PMC: LDS bank-conflict fraction 0.58 of this kernel).  Rows are padded
    // to SF bytes with SF / 4 odd, so that consecutive rows start in different banks; the ray block (6 floats per ray) to 7.
    __shared__ unsigned char s_f[TSDF_CHUNK + 64 * 8];
    __shared__ float s_ray[64 * 7];
    __shared__ int s_cnt, s_base;
    if (threadIdx.x == 0) s_cnt = 0;
    const int lane = threadIdx.x & 63;
    const bool rays = a.P.mode == ADFP_PTS_RAYS;
    const int S = a.P.S;
    const int SF = rays ? 4 * (((S + 3) >> 2) | 1) : 0;          // flag row stride in LDS (ray mode)
    int RB = 1, q0, npts, rb = 1;
    if (rays) {
        RB = a.rays_per_block;
        const int r0 = blockIdx.x * RB;
        const int nr = a.P.n / S;
        rb = (r0 + RB <= nr) ? RB : nr - r0;
        q0 = r0 * S; npts = rb * S;
        double zr[PER];                              // all of the thread's z loads in flight at once
#pragma unroll
        for (int k = 0; k < PER; ++k) { const int i = threadIdx.x + 256 * k; zr[k] = i < npts ? a.P.z[q0 + i] : 0.0; }
        float rr[2] = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < rb * 6) { const int row = i / 6, c = i - row * 6; rr[k] = c < 3 ? a.P.ro[3 * (r0 + row) + c] : a.P.rd[3 * (r0 + row) + c - 3]; }
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < npts) { const int row = (int)((unsigned)i / (unsigned)S), col = i - row * S; s_z[row * (S + 1) + col] = zr[k]; }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) { const int i = threadIdx.x + 256 * k; if (i < rb * 6) { const int row = i / 6; s_ray[row * 7 + (i - row * 6)] = rr[k]; } }
    } else {
        q0 = blockIdx.x * TSDF_CHUNK;
        npts = a.P.n - q0 < TSDF_CHUNK ? a.P.n - q0 : TSDF_CHUNK;
    }
    __syncthreads();
    const bool paired = a.t.sZ == 1 && a.t.Z >= 2;   // the reference's layout: z fastest
    for (int i0 = 0; i0 < npts; i0 += 256 * UB) {    // block-uniform trip count (ballots below)
        double p[UB][3]; float pn[UB][3]; int loc[UB], floc[UB]; bool ok[UB]; float tv[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int i = i0 + 256 * u + threadIdx.x;
            ok[u] = i < npts;
            const int ic = ok[u] ? i : npts - 1;     // clamped: every lane does valid loads, results masked
            loc[u] = ic; floc[u] = ic;
            if (rays) {
                const int col = (int)((unsigned)ic / (unsigned)rb), row = ic - col * rb;   // transposed walk: lane <-> ray
                loc[u] = row * S + col; floc[u] = row * SF + col;
                const double z = s_z[row * (S + 1) + col];
#pragma unroll
                for (int k = 0; k < 3; ++k) p[u][k] = __dadd_rn((double)s_ray[row * 7 + k], __dmul_rn((double)s_ray[row * 7 + 3 + k], z));
            } else load_point(a.P, q0 + ic, p[u]);
            normalize3(a.nt, p[u], pn[u]);
        }
        if (a.t.cb) {                                // the corner-block copy: one aligned 32-byte piece per point (adfp_device.h)
            TriBlock tb[UB]; f32x4 vl[UB], vh[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) trilerp_block_prepare(a.t, pn[u], tb[u]);
#pragma unroll
            for (int u = 0; u < UB; ++u) { vl[u] = tb[u].a[0]; vh[u] = tb[u].a[1]; }
#pragma unroll
            for (int u = 0; u < UB; ++u) tv[u] = trilerp_block_finish(tb[u], vl[u], vh[u]);
        } else if (paired) {
            TriPair tp[UB]; f32x2_u v[UB][4];
#pragma unroll
            for (int u = 0; u < UB; ++u) trilerp_pair_prepare(a.t, pn[u], tp[u]);
#pragma unroll
            for (int u = 0; u < UB; ++u) { v[u][0] = *tp[u].a00; v[u][1] = *tp[u].a01; v[u][2] = *tp[u].a10; v[u][3] = *tp[u].a11; }
#pragma unroll
            for (int u = 0; u < UB; ++u) tv[u] = trilerp_pair_finish(tp[u], v[u][0], v[u][1], v[u][2], v[u][3]);
        } else {
#pragma unroll
            for (int u = 0; u < UB; ++u) tv[u] = trilerp_scalar(a.t, pn[u]);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const float t = tv[u];
            const int q = q0 + loc[u];
            // a NaN position (degenerate ray, see k_sample) is in no band: F.grid_sample of a NaN coordinate is NaN on
            // the reference's side and every comparison with it is false (decoder.py:329)
            const bool pnan = (p[u][0] != p[u][0]) | (p[u][1] != p[u][1]) | (p[u][2] != p[u][2]);
            const bool band = ok[u] & !pnan & (t > (float)(-1.0 + 1e-4)) & (t < (float)(1.0 - 1e-4));
            if (ok[u]) {
                if (a.tsdf_out) a.tsdf_out[q] = t;
                s_f[floc[u]] = (unsigned char)((in_bound(p[u], a.b) ? ADFP_F_INBOUND : 0u) | (band ? ADFP_F_BAND : 0u));
            }
            if (a.list) {
                const unsigned long long m = __ballot(band);
                if (m) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&s_cnt, __popcll(m));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (band) {
                        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                        s_q[pos] = q; s_u[pos] = inv_tsdf(t);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (a.flags) {
        if (!rays) {
            if (((q0 | npts) & 3) == 0) {            // whole block 4-byte aligned: 4 flags per store
                for (int i = threadIdx.x; i < (npts >> 2); i += 256) ((unsigned*)(a.flags + q0))[i] = ((const unsigned*)s_f)[i];
            } else {
                for (int i = threadIdx.x; i < npts; i += 256) a.flags[q0 + i] = s_f[i];
            }
        } else if ((S & 3) == 0) {                   // rows are whole words: 4 flags per store, un-padding on the way out
            for (int i = threadIdx.x; i < (npts >> 2); i += 256) {
                const int e = 4 * i, row = (int)((unsigned)e / (unsigned)S), col = e - row * S;
                ((unsigned*)(a.flags + q0))[i] = *(const unsigned*)(s_f + row * SF + col);
            }
        } else {
            for (int i = threadIdx.x; i < npts; i += 256) { const int row = (int)((unsigned)i / (unsigned)S); a.flags[q0 + i] = s_f[row * SF + (i - row * S)]; }
        }
    }
    if (a.w) for (int i = threadIdx.x; i < npts; i += 256) a.w[q0 + i] = 1.f;
    if (!a.list) return;
    const int n = s_cnt;
    if (n == 0) return;
    if (threadIdx.x == 0) s_base = atomicAdd(a.counter, n);
    __syncthreads();
    const int base = s_base;
    for (int i = threadIdx.x; i < n; i += 256) { a.list[base + i] = s_q[i]; a.att_u[base + i] = s_u[i]; }
}

// =====================================================================================
// decoder kernels (a7-a9): gather -> Fourier -> 5 layers on MFMA -> output layer on VALU
// =====================================================================================
#define ROLE_LOW 0
#define ROLE_HIGH 1
#define ROLE_COLOR 2

struct DecodeArgs {
    PtsDev P; NormDev nb; double b[6];
    GridDev g0;                // own grid
    GridDev g1;                // low grid (HIGH only: concat_feature, decoder.py:182-187)
    const float* packed;
    const int* list;           // HIGH: in-band point ids
    const int* count_ptr;      // HIGH: device-side list length
    const unsigned char* flags;
    float* raw;                // [P,4]
    float* w;                  // [P] (LOW writes 1 when there is no TSDF stage)
    float* att_occ;            // HIGH: high+low per list entry
    int write_w;
    int apply_bound;           // Renderer.eval_points' ret[~mask,3] = 100 (Renderer.py:64)
    int* status;               // adfp_scene.status (f16x3 kernels: operand range guard) or NULL
    unsigned* masks;           // training forward (k_decode_h<..., 1>): ReLU masks, [rows][2][3] words
    float* act;                // training forward: X part of the staging rows, [rows][DecStage::NX], or NULL
    int single;                // adfp_decode_single: one decoder alone (COLOR writes its 4th output, HIGH does not add `low`)
    int* call_flag;            // f16x3 kernels: the call's range flag (device memory, arms k_fallback_points) or NULL
    int* pool = nullptr;       // k_decode_high_g: counter of the chip-wide tile tail (claim_tile_pool), zero at launch, or NULL
};

template <int CDIM, int NOUT, int ROLE, int NT>
__global__ __launch_bounds__(NT, 2) void k_decode(DecodeArgs a) {
    using L = DecLayout<CDIM, NOUT>;
    __shared__ __attribute__((aligned(16))) float lds[L::P_TOTAL];
    for (int i = threadIdx.x; i < L::P_TOTAL / 4; i += NT) ((f32x4*)lds)[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
    const int lane_off4 = h * ADFP_RG + p * 4;
    const int wave = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (NT / 64);
    const int count = (ROLE == ROLE_HIGH && a.count_ptr) ? *a.count_ptr : a.P.n;
    const int ntiles = (count + 31) >> 5;

    for (int tile = wave; tile < ntiles; tile += nwaves) {
        const int idx = tile * 32 + p;
        const bool valid = idx < count;
        int q = valid ? idx : 0;
        if (ROLE == ROLE_HIGH && a.list) q = a.list[q];

        double pt[3]; float pn[3], pf[3];
        load_point(a.P, q, pt);
        normalize3(a.nb, pt, pn);
        pf[0] = (float)pt[0]; pf[1] = (float)pt[1]; pf[2] = (float)pt[2];   // p.float() decoder.py:189

        float c[L::KSC];
        gather16(a.g0, pn, h, c);
        if (CDIM == 64) gather16(a.g1, pn, h, c + 16);

        // Fourier features sin(p @ B); k-step s carries feature unit_of(s, h) (decoder.py:26-30)
        float e[L::KSE];
#pragma unroll
        for (int s = 0; s < L::KSE; ++s) {
            const f32x4 bm = *(const f32x4*)(lds + L::P_BM + (unit_of(s, 0) + 4 * h) * 4);
            const float arg = fmaf(pf[2], bm.z, fmaf(pf[1], bm.y, pf[0] * bm.x));
            e[s] = adfp_sinf(arg);
        }

        // h = relu(W_i h + b_i) + (Wc_i c + bc_i); skip-concat [emb, h] feeds layer 3 (decoder.py:192-199)
        f32x16 hcur, acc;
        bias_init(acc, lds + L::P_BP(0), h);
        mfma_chain<L::KSE>(acc, lds + L::P_WP(0), lane_off4, e);
        relu_bias(acc, lds + L::P_BC(0), h);
        mfma_chain<L::KSC>(acc, lds + L::P_WC(0), lane_off4, c);
        hcur = acc;
#pragma unroll
        for (int i = 1; i < 5; ++i) {
            bias_init(acc, lds + L::P_BP(i), h);
            if (i == 3) {
                mfma_chain<L::KSE>(acc, lds + L::P_WP(i), lane_off4, e);
                mfma_chain<16>(acc, lds + L::P_WP(i) + L::chain_floats(L::KSE), lane_off4, hcur);
            } else {
                mfma_chain<16>(acc, lds + L::P_WP(i), lane_off4, hcur);
            }
            relu_bias(acc, lds + L::P_BC(i), h);
            mfma_chain<L::KSC>(acc, lds + L::P_WC(i), lane_off4, c);
            hcur = acc;
        }

        // output_linear on the VALU: each half holds 16 of the 32 hidden units
        float out[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const float* wo = lds + L::P_WO + (h * NOUT + o) * 16;
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s = fmaf(hcur[r], wo[r], s);
            s += __shfl_xor(s, 32);
            out[o] = s + lds[L::P_BO + o];
        }

        nan_point_outputs<NOUT>(pt, out);
        if (valid && h == 0) {
            if constexpr (ROLE == ROLE_LOW) {
                const bool inb = in_bound(pt, a.b);
                const unsigned f = a.flags ? a.flags[q] : 0u;
                // in-band points keep the true value for the HIGH pass; the attention pass
                // overwrites them (and applies the bound rule) afterwards.
                a.raw[4ll * q + 3] = ((f & ADFP_F_BAND) || inb || !a.apply_bound) ? out[0] : 100.f;   // Renderer.py:64
                if (a.write_w) a.w[q] = 1.f;
            } else if constexpr (ROLE == ROLE_COLOR) {
                a.raw[4ll * q + 0] = out[0]; a.raw[4ll * q + 1] = out[1]; a.raw[4ll * q + 2] = out[2];
                if (a.single) a.raw[4ll * q + 3] = out[3];
            } else {
                a.att_occ[idx] = a.single ? out[0] : out[0] + a.raw[4ll * q + 3];    // high + low, decoder.py:342
            }
        }
    }
}

struct AttArgs {
    const float* packed; const int* list; const int* count_ptr;
    const float* att_occ; const float* att_u; const unsigned char* flags;
    float* raw; float* w; int apply_bound;
    int n_rows;                // rows when count_ptr == NULL (adfp_attention_rows)
    int* status;
    int* call_flag;            // as DecodeArgs.call_flag
    unsigned* masks;           // training forward (k_attention_h<1>): ReLU masks + softmax weights, [rows][2][ATT_MASK_WORDS / 2]
    float* act;                // training forward: X piece of the staging rows ([rows][416]: inputs, h_0..h_3), or NULL
    int* pool = nullptr;       // k_attention_g: counter of the chip-wide tile tail (claim_tile_pool), zero at launch, or NULL
};

// workgroup shape of the dense f16x3 decoder kernels: 256 threads x 2 workgroups per CU, or one
// workgroup of 512 / 768 / 1024 threads per CU (2 / 3 / 4 waves per SIMD sharing one weight image)
#ifndef ADFP_DECH_NT
#define ADFP_DECH_NT 768
#endif
#define ADFP_DECH_WG (ADFP_DECH_NT == 256 ? 2 : 1)
#ifndef ADFP_HIGH_NT
#define ADFP_HIGH_NT 512
#endif
#define ADFP_LC_NT ADFP_DECH_NT
#ifndef ADFP_LCT_NT
#define ADFP_LCT_NT 512          // the training forward of the fused low + colour launch: 212 VGPRs (at 768 threads 64 spilled)
#endif
#include "adfp_decode_h.h"
#include "adfp_fallback.h"

// =====================================================================================
// attention fusion mlp_tsdf (a11) on the in-band list
// =====================================================================================

__global__ __launch_bounds__(512, 2) void k_attention(AttArgs a) {
    using A = AttLayout;
    __shared__ __attribute__((aligned(16))) float lds[A::P_TOTAL];
    for (int i = threadIdx.x; i < A::P_TOTAL / 4; i += 512) ((f32x4*)lds)[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
    const int lane_off4 = h * ADFP_RG + p * 4;
    const int wave = blockIdx.x * 8 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 8;
    const int count = a.count_ptr ? *a.count_ptr : a.n_rows;
    const int ntiles = (count + 31) >> 5;
    for (int tile = wave; tile < ntiles; tile += nwaves) {
        const int idx = tile * 32 + p;
        const bool valid = idx < count;
        const int ii = valid ? idx : 0;
        const float occ = a.att_occ[ii], u = a.att_u[ii];
        // layer 0 (2 -> 64) on the VALU; k-step s carries unit unit_of(s, h)
        float h0[32];
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const f32x4 t = *(const f32x4*)(lds + A::P_A0 + (unit_of(s, 0) + 4 * h) * 4);
            h0[s] = fmaxf(fmaf(u, t.y, fmaf(occ, t.x, t.z)), 0.f);
        }
        // layer 1: 64 -> 128
        float h1[64];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            f32x16 acc;
            bias_init(acc, lds + A::P_B1 + 32 * ob, h);
            mfma_chain<32>(acc, lds + A::P_W1 + ob * A::BLK1, lane_off4, h0);
#pragma unroll
            for (int r = 0; r < 16; ++r) h1[16 * ob + r] = fmaxf(acc[r], 0.f);
        }
        // layer 2: 128 -> 128
        float h2[64];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            f32x16 acc;
            bias_init(acc, lds + A::P_B2 + 32 * ob, h);
            mfma_chain<64>(acc, lds + A::P_W2 + ob * A::BLK2, lane_off4, h1);
#pragma unroll
            for (int r = 0; r < 16; ++r) h2[16 * ob + r] = fmaxf(acc[r], 0.f);
        }
        // layer 3: 128 -> 64, output 64 -> 2 on the VALU
        float l0 = 0.f, l1 = 0.f;
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            f32x16 acc;
            bias_init(acc, lds + A::P_B3 + 32 * ob, h);
            mfma_chain<64>(acc, lds + A::P_W3 + ob * A::BLK2, lane_off4, h2);
            const float* w0 = lds + A::P_WO + (h * 2 + 0) * 32 + 16 * ob;
            const float* w1 = lds + A::P_WO + (h * 2 + 1) * 32 + 16 * ob;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = fmaxf(acc[r], 0.f);
                l0 = fmaf(v, w0[r], l0);
                l1 = fmaf(v, w1[r], l1);
            }
        }
        l0 += __shfl_xor(l0, 32); l1 += __shfl_xor(l1, 32);
        l0 += lds[A::P_BO]; l1 += lds[A::P_BO + 1];
        // softmax over 2, convex blend (decoder.py:255-258)
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        const float den = e0 + e1;
        const float a0 = e0 / den, a1 = e1 / den;
        const float fused = a0 * occ + a1 * u;
        if (valid && h == 0) {
            const int q = a.list ? a.list[ii] : ii;
            const bool inb = !a.flags || (a.flags[q] & ADFP_F_INBOUND) != 0;
            a.raw[4ll * q + 3] = (inb || !a.apply_bound) ? fused : 100.f;   // Renderer.py:64
            a.w[q] = a1;
        }
    }
}

// =====================================================================================
// a13: compositing.  One wave per ray, lane = sample (chunks of 64), wave scan for T.
// =====================================================================================
__global__ __launch_bounds__(256) void k_composite(const float* __restrict__ raw, const double* __restrict__ z, int n_rays, int S,
                                                   double* __restrict__ depth, double* __restrict__ var,
                                                   float* __restrict__ color, float* __restrict__ weights) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    float carry = 1.f;                 // running prod of (1 - alpha + 1e-10) over previous chunks
    float cr = 0.f, cg = 0.f, cb = 0.f;
    double sw = 0.0, swz = 0.0, swzz = 0.0;
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        const bool ok = s < S;
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        double zz = 0.0;
        if (ok) { r = *(const f32x4*)(raw + ((long long)ray * S + s) * 4); zz = z[(long long)ray * S + s]; }
        const float alpha = ok ? sigmoidf_(10.f * r.w) : 0.f;                   // common.py:236
        float f = ok ? (1.f - alpha + 1e-10f) : 1.f;
        // inclusive product scan across the wave (DPP), then one lane to the right for the exclusive one
        float total;
        const float incl = wave_scan_mul(f, lane, total);
        const float excl = dpp_f32<0x138, false>(incl, 1.f);                       // wave_shr:1, lane 0 <- 1
        const float T = carry * excl;
        const float w = alpha * T;                                                // common.py:244
        carry *= total;
        if (ok && weights) weights[(long long)ray * S + s] = w;
        cr = fmaf(w, r.x, cr); cg = fmaf(w, r.y, cg); cb = fmaf(w, r.z, cb);
        const double wd = (double)w;
        sw += wd; swz += wd * zz; swzz += wd * zz * zz;
    }
    cr = wave_sum(cr); cg = wave_sum(cg); cb = wave_sum(cb);
    sw = wave_sum(sw); swz = wave_sum(swz); swzz = wave_sum(swzz);
    if (lane == 0) {
        depth[ray] = swz;
        // sum w (z - d)^2 = sum w z^2 - 2 d sum w z + d^2 sum w        (common.py:248-250)
        var[ray] = swzz - 2.0 * swz * swz + swz * swz * sw;
        color[3 * ray + 0] = cr; color[3 * ray + 1] = cg; color[3 * ray + 2] = cb;
    }
}

#include "adfp_sort.h"
#include "adfp_backward.h"
#include "adfp_backward_h.h"
#include "adfp_backward_fused.h"
#include "adfp_backward_roles.h"
#ifdef ADFP_STAMPS_ROLES
extern "C" int adfp_debug_roles_span(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_roles_span), sizeof(unsigned long long) * 4 * 256);
}
#endif
#ifdef ADFP_STAMPS
extern "C" int adfp_debug_phases_fused(unsigned long long* host_out, int reset) {
    int rc = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase_fused), 64);
    if (!rc && reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_fused), z, 64); }
    return rc;
}
#endif
#include "adfp_fusion.h"
#include "adfp_mapping.h"
#include "adfp_mapper_iter.h"
#include "adfp_tracker_iter.h"
#include "adfp_decode_g.h"
#ifdef ADFP_STAMPS_G
extern "C" int adfp_debug_phases_g(unsigned long long* host_out, int reset) {
    int rc = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase_g), 192);
    if (!rc && reset) { unsigned long long z[24] = {}; rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_g), z, 192); }
    return rc;
}
extern "C" int adfp_debug_wave_span_g(unsigned long long* host_out, int n_waves) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wave_span_g), (size_t)n_waves * 16);
}
#endif

// adfp_ray_sort_keys (adfp.h): Morton keys of (origin cell, surface-point cell) per ray
__device__ __forceinline__ unsigned morton3(unsigned x, unsigned y, unsigned z, int bits) {
    unsigned k = 0;
    for (int b = 0; b < bits; ++b) k |= (((x >> b) & 1u) << (3 * b)) | (((y >> b) & 1u) << (3 * b + 1)) | (((z >> b) & 1u) << (3 * b + 2));
    return k;
}
struct RayKeyArgs { const float* ro; const float* rd; const float* gd; int n; float lo[3], inv[3]; int* key; int* val; };
__global__ __launch_bounds__(256) void k_ray_sort_keys(RayKeyArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    float t = a.gd ? a.gd[i] : 1.f;
    if (!(t > 0.f) || !(t < 3.0e38f)) t = 1.f;
    unsigned co[3], cs[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float o = a.ro[3 * i + k], p = fmaf(a.rd[3 * i + k], t, o);
        const float uo = (o - a.lo[k]) * a.inv[k], us = (p - a.lo[k]) * a.inv[k];
        co[k] = (unsigned)fminf(fmaxf(uo * 4.f, 0.f), 3.f);           // NaN -> 0
        cs[k] = (unsigned)fminf(fmaxf(us * 256.f, 0.f), 255.f);
    }
    a.key[i] = (int)((morton3(co[0], co[1], co[2], 2) << 24) | morton3(cs[0], cs[1], cs[2], 8));
    a.val[i] = i;
}

// adfp_ray_order_probe (adfp.h): one workgroup, 2048 pairs of consecutive rays
__global__ __launch_bounds__(256) void k_ray_order_probe(const float* __restrict__ ro, const float* __restrict__ rd, const float* __restrict__ gd, int n,
                                                         float far2, int* __restrict__ verdict) {
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int pairs = n - 1 < 2048 ? n - 1 : 2048;
    const int stride = pairs > 0 ? (n - 1) / pairs : 1;
    int far = 0;
    for (int k = threadIdx.x; k < pairs; k += 256) {
        const int i = k * stride;
        float p[2][3];
        for (int e = 0; e < 2; ++e) {
            float t = gd ? gd[i + e] : 1.f;
            if (!(t > 0.f) || !(t < 3.0e38f)) t = 1.f;
            for (int c = 0; c < 3; ++c) p[e][c] = fmaf(rd[3 * (i + e) + c], t, ro[3 * (i + e) + c]);
        }
        const float dx = p[1][0] - p[0][0], dy = p[1][1] - p[0][1], dz = p[1][2] - p[0][2];
        far += (dx * dx + dy * dy + dz * dz > far2) ? 1 : 0;
    }
    atomicAdd(&s_cnt, far);
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(verdict + 1, pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(verdict, s_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The two sides of a sharded render's ONE all-gather (adfp.h: adfp_gather_pack / adfp_gather_unpack): several per-ray arrays <->
// one interleaved row buffer.  blockIdx.y = rank (unpack), a thread moves one 4-byte word.
struct GatherJobs { unsigned* arr[ADFP_GATHER_MAX]; int words[ADFP_GATHER_MAX]; int woff[ADFP_GATHER_MAX + 1]; int n, world; long long pad; long long prefix[ADFP_GATHER_MAX_RANKS + 1]; };
template <bool PACK>
__global__ __launch_bounds__(256) void k_gather_rows(GatherJobs j, unsigned* __restrict__ buf) {
    const int W = j.woff[j.n];
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int rank = blockIdx.y;
    const long long row = t / W;
    const int word = (int)(t - row * W);
    if (row >= j.prefix[rank + 1] - j.prefix[rank]) return;
    int a = 0;
    while (a + 1 < j.n && word >= j.woff[a + 1]) ++a;
    unsigned* arr = j.arr[a] + (j.prefix[rank] + row) * j.words[a] + (word - j.woff[a]);
    unsigned* slot = buf + ((long long)rank * j.pad + row) * W + word;
    if (PACK) *slot = *arr; else *arr = *slot;
}

// =====================================================================================
// host side: C ABI
// =====================================================================================
// compute units of the CURRENT device (launch geometry of the persistent kernels).  Asked per call: the library keeps no global
// mutable state, and a process may drive GPUs of different sizes.  hipDeviceGetAttribute is a table lookup (no device round trip).
// t_cu_reserve: compute units the calling thread's CURRENT entry point leaves to its side lane (adfp_backward_args.side_stream): the
// persistent kernels of the backward take a whole CU per workgroup (132-160 KB of LDS, the whole register file), so the sort's
// short launches on the second stream would otherwise wait for a whole kernel to retire (measured: a 6-us histogram launch took
// 130 us behind k_decode_bwd_roles).  Set and restored by backward_points (CuReserve); 0 everywhere else.
static thread_local int t_cu_reserve = 0;
#ifndef ADFP_SIDE_CU_RESERVE
#define ADFP_SIDE_CU_RESERVE 4
#endif
static int num_cu() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    n -= t_cu_reserve;
    return n < 1 ? 1 : n;
}

static NormDev make_norm(const double b[3][2]) {
    NormDev n;
    for (int k = 0; k < 3; ++k) { n.lo[k] = b[k][0]; n.inv[k] = 1.0 / (b[k][1] - b[k][0]); }
    return n;
}
static void fill_bound(double out[6], const double b[3][2]) {
    for (int k = 0; k < 3; ++k) { out[2 * k] = b[k][0]; out[2 * k + 1] = b[k][1]; }
}
// gather16 addresses a grid with 32-bit byte offsets: 16.7 M voxels (the reference's largest is 0.3 M)
static bool grid_too_big(const adfp_grid& g) { return g.data && (long long)g.Z * g.Y * g.X * 128 >= (1ll << 31); }
static GridDev make_grid(const adfp_grid& g) {
    GridDev d; d.data = g.data; d.Z = g.Z; d.Y = g.Y; d.X = g.X;
    d.fZ1 = (float)(g.Z - 1); d.fY1 = (float)(g.Y - 1); d.fX1 = (float)(g.X - 1);
    return d;
}
static TsdfDev make_tsdf(const adfp_tsdf& t) {
    TsdfDev d; d.data = t.data; d.Z = t.Z; d.Y = t.Y; d.X = t.X; d.sZ = t.sZ; d.sY = t.sY; d.sX = t.sX; d.cb = t.corner_blocks; return d;
}
static int make_pts(const adfp_points* p, PtsDev* d) {
    if (!p || p->n_points < 0) return ADFP_E_ARG;
    if (p->n_points > 0x7fffffffll) return ADFP_E_UNSUPPORTED;
    d->mode = p->mode; d->S = p->S > 0 ? p->S : 1; d->n = (int)p->n_points;
    d->pts = p->pts; d->ro = p->rays_o; d->rd = p->rays_d; d->z = p->z_vals;
    if (p->mode == ADFP_PTS_RAYS) { if (!p->rays_o || !p->rays_d || !p->z_vals || p->S <= 0) return ADFP_E_ARG; }
    else if (p->mode == ADFP_PTS_F64 || p->mode == ADFP_PTS_F32) { if (!p->pts && p->n_points) return ADFP_E_ARG; }
    else return ADFP_E_ARG;
    return 0;
}

// workspace carve-up (all offsets 256-B aligned)
struct Workspace {
    double* z; float* raw; unsigned char* flags; int* list; float* att_occ; float* att_u; int* counter;
    unsigned* segparts;          // [SEGMAX_SEGS][SEGMAX_PARTS] ordered-uint partial maxima of gt_depth (k_forward_head -> k_sample)
    size_t bytes;
};
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
// (offsets are formed as integers: the size queries carve a NULL base, and pointer arithmetic on NULL is undefined)
template <typename T> static T* at(void* base, size_t o) { return (T*)((uintptr_t)base + o); }
static Workspace carve(void* base, long long P) {
    Workspace w; size_t o = 0;
    w.counter = at<int>(base, o); o += 256;
    w.segparts = at<unsigned>(base, o); o += align256((size_t)SEGMAX_SEGS * SEGMAX_PARTS * 4);
    w.z = at<double>(base, o); o += align256((size_t)P * 8);
    w.raw = at<float>(base, o); o += align256((size_t)P * 16);
    w.list = at<int>(base, o); o += align256((size_t)P * 4);
    w.att_occ = at<float>(base, o); o += align256((size_t)P * 4);
    w.att_u = at<float>(base, o); o += align256((size_t)P * 4);
    w.flags = at<unsigned char>(base, o); o += align256((size_t)P);
    w.bytes = o;
    return w;
}

// Several images in ONE launch: a training iteration re-packs four per step (the trained networks' forward and transposed images) and
// a pack kernel is ~5 us inside a graph replay whatever it packs.  A workgroup belongs to one job (its first block is in `first`).
struct PackJobs { int n; int first[ADFP_PACK_MAX_JOBS + 1]; int net[ADFP_PACK_MAX_JOBS]; int fmt[ADFP_PACK_MAX_JOBS];
                  const float* flat[ADFP_PACK_MAX_JOBS]; unsigned* out[ADFP_PACK_MAX_JOBS]; int* status; };
static int pack_job_blocks(int net, int fmt) {
    const bool h = fmt == ADFP_IMAGE_H, g = fmt == ADFP_IMAGE_G, t = fmt == ADFP_IMAGE_HT;
    switch (net) {
        case ADFP_DEC_LOW: return h ? DecLayoutH<32, 1>::NFLAG : (g ? DecLayoutG<32, 1>::NFLAG : (t ? (DecLayoutHT<32, 1>::P_TOTAL + 255) / 256 : -1));
        case ADFP_DEC_HIGH: return h ? DecLayoutH<64, 1>::NFLAG : (g ? DecLayoutG<64, 1>::NFLAG : (t ? (DecLayoutHT<64, 1>::P_TOTAL + 255) / 256 : -1));
        case ADFP_DEC_COLOR: return h ? DecLayoutH<32, 4>::NFLAG : (g ? DecLayoutG<32, 4>::NFLAG : (t ? (DecLayoutHT<32, 4>::P_TOTAL + 255) / 256 : -1));
        case ADFP_NET_ATT: return h ? AttLayoutH::NFLAG : (g ? AttLayoutG::NFLAG : (t ? (AttLayoutHT::P_TOTAL + 255) / 256 : -1));
    }
    return -1;
}
template <int CDIM, int NOUT>
ADFP_DEV void pack_decoder_any(int fmt, int blk, const float* flat, unsigned* out, int* status, int bit) {
    if (fmt == ADFP_IMAGE_H) pack_decoder_h_block<CDIM, NOUT>(blk, flat, out, status, bit);
    else if (fmt == ADFP_IMAGE_G) pack_decoder_g_block<CDIM, NOUT>(blk, flat, out + DecLayoutH<CDIM, NOUT>::P_TOTAL, status, bit);      // the G part lies behind the H part
    else pack_decoder_ht_block<CDIM, NOUT>(blk, flat, out, status, bit);
}
__device__ inline void pack_multi_block(const PackJobs& j, int block) {
    int k = 0;
    while (k + 1 < j.n && block >= j.first[k + 1]) ++k;                     // block-uniform
    const int blk = block - j.first[k], fmt = j.fmt[k];
    const float* flat = j.flat[k];
    unsigned* out = j.out[k];
    switch (j.net[k]) {
        case ADFP_DEC_LOW: pack_decoder_any<32, 1>(fmt, blk, flat, out, j.status, ADFP_STATUS_F16_RANGE_LOW); break;
        case ADFP_DEC_HIGH: pack_decoder_any<64, 1>(fmt, blk, flat, out, j.status, ADFP_STATUS_F16_RANGE_HIGH); break;
        case ADFP_DEC_COLOR: pack_decoder_any<32, 4>(fmt, blk, flat, out, j.status, ADFP_STATUS_F16_RANGE_COLOR); break;
        default:
            if (fmt == ADFP_IMAGE_H) pack_attention_h_block(blk, flat, out, j.status);
            else if (fmt == ADFP_IMAGE_G) pack_attention_g_block(blk, flat, out + AttLayoutH::P_TOTAL, j.status);
            else pack_attention_ht_block(blk, flat, out, j.status);
    }
}
__global__ __launch_bounds__(256) void k_pack_multi(PackJobs j) { pack_multi_block(j, (int)blockIdx.x); }
// the first launch of a render call: the weight images it was handed to pack (adfp_render_args.pack_jobs) and the zero fill of its
// device words
// ... and, since round 6, two more kinds of blocks, so that a frame (or a rank's share of one) needs no launch of its own for either:
//   rays:   the rays of pixels [first, first + n) of an H x W frame (k_get_rays' arithmetic, src/common.py:254-272)
//   segmax: SEGMAX_PARTS partial maxima of gt_depth per segment of `seg` rays, as ordered uints, plain stores (no atomics: nothing
//           has to be zeroed before this launch); k_sample's lanes fold them.  Block (s, p) takes the p-th sixteenth of segment s.
struct RayJob { const float* c2w; int W; float fx, fy, cx, cy; int first, n; float* ro; float* rd; };
struct SegJob { const float* d; int n, seg, nseg; unsigned* parts;
                // pre-filter mode (adfp_render_args.prefilter_bound): the maximum is taken over the rays the Mapper's bounding-box test keeps,
                // and the test's verdict is written to keep[ray] -- k_prefilter_mask's arithmetic (adfp_mapper_iter.h)
                const float* ro; const float* rd; const double* bnd; unsigned char* keep; };
struct ForwardHeadArgs { PackJobs p; ZeroJobs z; int nb_pack, nb_zero, nb_rays; RayJob rj; SegJob sj; };
__device__ inline void rays_block(const RayJob& j, unsigned blk) {
    const int k = (int)blk * 256 + threadIdx.x;
    if (k >= j.n) return;
    const int idx = j.first + k;
    const int row = idx / j.W, col = idx - row * j.W;
    const float dx = ((float)col - j.cx) / j.fx, dy = -((float)row - j.cy) / j.fy, dz = -1.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float s = __fadd_rn(__fadd_rn(__fmul_rn(dx, j.c2w[4 * a + 0]), __fmul_rn(dy, j.c2w[4 * a + 1])), __fmul_rn(dz, j.c2w[4 * a + 2]));
        j.rd[3 * k + a] = s;
        j.ro[3 * k + a] = j.c2w[4 * a + 3];
    }
}
__device__ inline void segmax_block(const SegJob& j, unsigned blk) {
    __shared__ unsigned s_m[4];
    const int sg = (int)blk / SEGMAX_PARTS, part = (int)blk % SEGMAX_PARTS;
    const long long s_lo = (long long)sg * j.seg, s_hi = s_lo + j.seg < j.n ? s_lo + j.seg : j.n;
    const long long len = s_hi - s_lo, per = (len + SEGMAX_PARTS - 1) / SEGMAX_PARTS;
    const long long lo = s_lo + part * per, hi = lo + per < s_hi ? lo + per : s_hi;
    unsigned m = 0;
    if (j.keep) {                                      // block-uniform
        double b[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) b[k] = j.bnd[k];
        float mx = -INFINITY;                           // k_prefilter_mask's start value: no kept ray -> -inf
        for (long long i = lo + threadIdx.x; i < hi; i += 256) {
            double t = INFINITY; bool nan = false;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double o = (double)j.ro[3 * i + k], d = (double)j.rd[3 * i + k];
                const double t0 = (b[2 * k] - o) / d, t1 = (b[2 * k + 1] - o) / d;
                nan |= (t0 != t0) | (t1 != t1);        // torch.max / torch.min propagate NaN
                const double tm = t0 > t1 ? t0 : t1;
                t = tm < t ? tm : t;
            }
            const float dep = j.d[i];
            const bool k_ = !nan && (t >= (double)dep);
            j.keep[i] = k_ ? 1 : 0;
            if (k_) mx = dep > mx ? dep : mx;
        }
        m = f2ord(mx);
    } else
    // eight independent loads per thread and trip: the slice is a few thousand floats, the cost is the latency of the trips
    for (long long i = lo + threadIdx.x; i < hi; i += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = i + 256 * u < hi ? j.d[i + 256 * u] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) { const unsigned q = i + 256 * u < hi ? f2ord(v[u]) : 0u; m = q > m ? q : m; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned q = __shfl_xor(m, o); m = q > m ? q : m; }
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) m = s_m[w] > m ? s_m[w] : m;
        j.parts[blk] = m;                                   // 0 = "below every float": an empty slice
    }
}
__global__ __launch_bounds__(256) void k_forward_head(ForwardHeadArgs h) {
    unsigned b = blockIdx.x;
    if ((int)b < h.nb_pack) { pack_multi_block(h.p, (int)b); return; }
    b -= (unsigned)h.nb_pack;
    if ((int)b < h.nb_zero) { zero_multi_block(h.z, b); return; }
    b -= (unsigned)h.nb_zero;
    if ((int)b < h.nb_rays) { rays_block(h.rj, b); return; }
    segmax_block(h.sj, b - (unsigned)h.nb_rays);
}
static int pack_jobs_table(int n_jobs, const adfp_pack_job* jobs, int* status, PackJobs& j) {
    if (n_jobs < 0 || n_jobs > ADFP_PACK_MAX_JOBS || (n_jobs && !jobs)) return ADFP_E_ARG;
    j.n = n_jobs; j.status = status; j.first[0] = 0;
    for (int k = 0; k < n_jobs; ++k) {
        const int nb = pack_job_blocks(jobs[k].net, jobs[k].format);
        if (nb <= 0 || !jobs[k].flat || !jobs[k].packed) return ADFP_E_ARG;
        j.net[k] = jobs[k].net; j.fmt[k] = jobs[k].format; j.flat[k] = jobs[k].flat; j.out[k] = (unsigned*)jobs[k].packed;
        j.first[k + 1] = j.first[k] + nb;
    }
    return 0;
}
extern "C" {

int adfp_version(void) { return ADFP_VERSION; }

long long adfp_decoder_flat_floats(int kind) {
    switch (kind) {
        case ADFP_DEC_LOW: return DecLayout<32, 1>::F_TOTAL;
        case ADFP_DEC_HIGH: return DecLayout<64, 1>::F_TOTAL;
        case ADFP_DEC_COLOR: return DecLayout<32, 4>::F_TOTAL;
    }
    return ADFP_E_ARG;
}
long long adfp_decoder_packed_floats(int kind) {
    switch (kind) {
        case ADFP_DEC_LOW: return DecLayout<32, 1>::P_TOTAL;
        case ADFP_DEC_HIGH: return DecLayout<64, 1>::P_TOTAL;
        case ADFP_DEC_COLOR: return DecLayout<32, 4>::P_TOTAL;
    }
    return ADFP_E_ARG;
}
long long adfp_attention_flat_floats(void) { return AttLayout::F_TOTAL; }
long long adfp_attention_packed_floats(void) { return AttLayout::P_TOTAL; }

size_t adfp_workspace_bytes(long long n_points) {
    if (n_points < 0) return 0;
    return carve(nullptr, n_points).bytes;
}

int adfp_relayout_grid(const float* src, float* dst, int C, int Z, int Y, int X, void* stream) {
    if (!src || !dst || Z <= 0 || Y <= 0 || X <= 0) return ADFP_E_ARG;
    if (C != 32) return ADFP_E_UNSUPPORTED;
    const long long V = (long long)Z * Y * X;
    hipLaunchKernelGGL(k_relayout_cm_to_cl, dim3((unsigned)((V + 63) / 64)), dim3(256), 0, (hipStream_t)stream, src, dst, V);
    ADFP_CHECK_LAUNCH();
    return 0;
}
static int relayout_jobs_table(int n_jobs, const adfp_relayout_job* jobs, RelayoutJobs& j) {
    if (n_jobs < 0 || n_jobs > ADFP_RELAYOUT_MAX_JOBS || (n_jobs && !jobs)) return ADFP_E_ARG;
    j.n = n_jobs; j.first[0] = 0;
    for (int k = 0; k < n_jobs; ++k) {
        if (!jobs[k].src || !jobs[k].dst || jobs[k].voxels <= 0) return ADFP_E_ARG;
        const long long nb = (jobs[k].voxels + 63) / 64;
        if (nb + j.first[k] > 0x7fffffffll) return ADFP_E_UNSUPPORTED;
        j.src[k] = jobs[k].src; j.dst[k] = jobs[k].dst; j.V[k] = jobs[k].voxels;
        j.first[k + 1] = j.first[k] + (unsigned)nb;
    }
    return 0;
}
int adfp_relayout_grids(int n_jobs, const adfp_relayout_job* jobs, int back, void* stream) {
    RelayoutJobs j;
    int rc = relayout_jobs_table(n_jobs, jobs, j); if (rc) return rc;
    if (n_jobs == 0) return 0;
    if (back) hipLaunchKernelGGL(k_relayout_multi<true>, dim3(j.first[j.n]), dim3(256), 0, (hipStream_t)stream, j);
    else hipLaunchKernelGGL(k_relayout_multi<false>, dim3(j.first[j.n]), dim3(256), 0, (hipStream_t)stream, j);
    ADFP_CHECK_LAUNCH();
    return 0;
}
int adfp_relayout_grid_back(const float* src, float* dst, int C, int Z, int Y, int X, void* stream) {
    if (!src || !dst || Z <= 0 || Y <= 0 || X <= 0) return ADFP_E_ARG;
    if (C != 32) return ADFP_E_UNSUPPORTED;
    const long long V = (long long)Z * Y * X;
    hipLaunchKernelGGL(k_relayout_cl_to_cm, dim3((unsigned)((V + 63) / 64)), dim3(256), 0, (hipStream_t)stream, src, dst, V);
    ADFP_CHECK_LAUNCH();
    return 0;
}

int adfp_pack_decoder(int kind, const float* flat, float* packed, void* stream) {
    if (!flat || !packed) return ADFP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (kind) {
        case ADFP_DEC_LOW:
            hipLaunchKernelGGL((k_pack_decoder<32, 1>), dim3((DecLayout<32, 1>::P_TOTAL + 255) / 256), dim3(256), 0, st, flat, packed);
            break;
        case ADFP_DEC_HIGH:
            hipLaunchKernelGGL((k_pack_decoder<64, 1>), dim3((DecLayout<64, 1>::P_TOTAL + 255) / 256), dim3(256), 0, st, flat, packed);
            break;
        case ADFP_DEC_COLOR:
            hipLaunchKernelGGL((k_pack_decoder<32, 4>), dim3((DecLayout<32, 4>::P_TOTAL + 255) / 256), dim3(256), 0, st, flat, packed);
            break;
        default: return ADFP_E_ARG;
    }
    ADFP_CHECK_LAUNCH();
    return 0;
}
// The split image of a 32-channel decoder is TWO images back to back: the "H" image (32x32x16 operand order: the training forward,
// the single-network kernels) and the "G" image (16x16x32 operand order, adfp_decode_g.h: the fused low + colour inference launch).
long long adfp_decoder_packed_h_words(int kind) {
    switch (kind) {
        case ADFP_DEC_LOW: return DecLayoutH<32, 1>::P_TOTAL + DecLayoutG<32, 1>::P_TOTAL;
        case ADFP_DEC_HIGH: return DecLayoutH<64, 1>::P_TOTAL + DecLayoutG<64, 1>::P_TOTAL;
        case ADFP_DEC_COLOR: return DecLayoutH<32, 4>::P_TOTAL + DecLayoutG<32, 4>::P_TOTAL;
    }
    return ADFP_E_ARG;
}
int adfp_pack_split_image(int net, int which, const float* flat, void* packed, int* status, void* stream) {
    if (!flat || !packed || !(which & (ADFP_IMAGE_H | ADFP_IMAGE_G)) || (which & ~(ADFP_IMAGE_H | ADFP_IMAGE_G))) return ADFP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    unsigned* out = (unsigned*)packed;
    const bool H = which & ADFP_IMAGE_H, G = which & ADFP_IMAGE_G;
    switch (net) {
        case ADFP_DEC_LOW:
            if (H) hipLaunchKernelGGL((k_pack_decoder_h<32, 1>), dim3(DecLayoutH<32, 1>::NFLAG), dim3(256), 0, st, flat, out, status, ADFP_STATUS_F16_RANGE_LOW);
            if (G) hipLaunchKernelGGL((k_pack_decoder_g<32, 1>), dim3(DecLayoutG<32, 1>::NFLAG), dim3(256), 0, st, flat, out + DecLayoutH<32, 1>::P_TOTAL, status, ADFP_STATUS_F16_RANGE_LOW);
            break;
        case ADFP_DEC_HIGH:
            if (H) hipLaunchKernelGGL((k_pack_decoder_h<64, 1>), dim3(DecLayoutH<64, 1>::NFLAG), dim3(256), 0, st, flat, out, status, ADFP_STATUS_F16_RANGE_HIGH);
            if (G) hipLaunchKernelGGL((k_pack_decoder_g<64, 1>), dim3(DecLayoutG<64, 1>::NFLAG), dim3(256), 0, st, flat, out + DecLayoutH<64, 1>::P_TOTAL, status, ADFP_STATUS_F16_RANGE_HIGH);
            break;
        case ADFP_DEC_COLOR:
            if (H) hipLaunchKernelGGL((k_pack_decoder_h<32, 4>), dim3(DecLayoutH<32, 4>::NFLAG), dim3(256), 0, st, flat, out, status, ADFP_STATUS_F16_RANGE_COLOR);
            if (G) hipLaunchKernelGGL((k_pack_decoder_g<32, 4>), dim3(DecLayoutG<32, 4>::NFLAG), dim3(256), 0, st, flat, out + DecLayoutH<32, 4>::P_TOTAL, status, ADFP_STATUS_F16_RANGE_COLOR);
            break;
        case ADFP_NET_ATT:
            if (H) hipLaunchKernelGGL(k_pack_attention_h, dim3(AttLayoutH::NFLAG), dim3(256), 0, st, flat, out, status);
            if (G) hipLaunchKernelGGL(k_pack_attention_g, dim3(AttLayoutG::NFLAG), dim3(256), 0, st, flat, out + AttLayoutH::P_TOTAL, status);
            break;
        default: return ADFP_E_ARG;
    }
    ADFP_CHECK_LAUNCH();
    return 0;
}
int adfp_pack_images(int n_jobs, const adfp_pack_job* jobs, int* status, void* stream) {
    PackJobs j;
    int rc = pack_jobs_table(n_jobs, jobs, status, j); if (rc) return rc;
    if (n_jobs == 0) return 0;
    hipLaunchKernelGGL(k_pack_multi, dim3(j.first[n_jobs]), dim3(256), 0, (hipStream_t)stream, j);
    ADFP_CHECK_LAUNCH();
    return 0;
}
int adfp_pack_decoder_h(int kind, const float* flat, void* packed, int* status, void* stream) {
    if (kind < ADFP_DEC_LOW || kind > ADFP_DEC_COLOR) return ADFP_E_ARG;
    return adfp_pack_split_image(kind, ADFP_IMAGE_H | ADFP_IMAGE_G, flat, packed, status, stream);
}
long long adfp_decoder_packed_ht_words(int kind) {
    switch (kind) {
        case ADFP_DEC_LOW: return DecLayoutHT<32, 1>::P_TOTAL;
        case ADFP_DEC_HIGH: return DecLayoutHT<64, 1>::P_TOTAL;
        case ADFP_DEC_COLOR: return DecLayoutHT<32, 4>::P_TOTAL;
    }
    return ADFP_E_ARG;
}
int adfp_pack_decoder_ht(int kind, const float* flat, void* packed, int* status, void* stream) {
    if (!flat || !packed) return ADFP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    unsigned* out = (unsigned*)packed;
    switch (kind) {
        case ADFP_DEC_LOW:
            hipLaunchKernelGGL((k_pack_decoder_ht<32, 1>), dim3((DecLayoutHT<32, 1>::P_TOTAL + 255) / 256), dim3(256), 0, st, flat, out, status, ADFP_STATUS_F16_RANGE_LOW);
            break;
        case ADFP_DEC_HIGH:
            hipLaunchKernelGGL((k_pack_decoder_ht<64, 1>), dim3((DecLayoutHT<64, 1>::P_TOTAL + 255) / 256), dim3(256), 0, st, flat, out, status, ADFP_STATUS_F16_RANGE_HIGH);
            break;
        case ADFP_DEC_COLOR:
            hipLaunchKernelGGL((k_pack_decoder_ht<32, 4>), dim3((DecLayoutHT<32, 4>::P_TOTAL + 255) / 256), dim3(256), 0, st, flat, out, status, ADFP_STATUS_F16_RANGE_COLOR);
            break;
        default: return ADFP_E_ARG;
    }
    ADFP_CHECK_LAUNCH();
    return 0;
}
long long adfp_train_act_floats(int kind) {
    switch (kind) {
        case ADFP_DEC_LOW: case ADFP_DEC_COLOR: return DecStage<32>::NXM;
        case ADFP_DEC_HIGH: return DecStage<64>::NXM;
    }
    return ADFP_E_ARG;
}
long long adfp_attention_packed_h_words(void) { return AttLayoutH::P_TOTAL + AttLayoutG::P_TOTAL; }      // H image, then G image (see adfp_decoder_packed_h_words)
int adfp_pack_attention_h(const float* flat, void* packed, int* status, void* stream) {
    return adfp_pack_split_image(ADFP_NET_ATT, ADFP_IMAGE_H | ADFP_IMAGE_G, flat, packed, status, stream);
}
long long adfp_attention_packed_ht_words(void) { return AttLayoutHT::P_TOTAL; }
int adfp_pack_attention_ht(const float* flat, void* packed, int* status, void* stream) {
    if (!flat || !packed) return ADFP_E_ARG;
    hipLaunchKernelGGL(k_pack_attention_ht, dim3((AttLayoutHT::P_TOTAL + 255) / 256), dim3(256), 0, (hipStream_t)stream, flat, (unsigned*)packed, status);
    ADFP_CHECK_LAUNCH();
    return 0;
}
int adfp_pack_attention(const float* flat, float* packed, void* stream) {
    if (!flat || !packed) return ADFP_E_ARG;
    hipLaunchKernelGGL(k_pack_attention, dim3((AttLayout::P_TOTAL + 255) / 256), dim3(256), 0, (hipStream_t)stream, flat, packed);
    ADFP_CHECK_LAUNCH();
    return 0;
}

int adfp_get_rays(int H, int W, float fx, float fy, float cx, float cy, const float* c2w, float* rays_o, float* rays_d, void* stream) {
    if (!c2w || !rays_o || !rays_d || H <= 0 || W <= 0) return ADFP_E_ARG;
    hipLaunchKernelGGL(k_get_rays, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, H, W, fx, fy, cx, cy, c2w, rays_o, rays_d);
    ADFP_CHECK_LAUNCH();
    return 0;
}

int adfp_rays_from_uv(const float* pix_i, const float* pix_j, int n, float fx, float fy, float cx, float cy, const float* c2w,
                      float* rays_o, float* rays_d, void* stream) {
    if (n < 0 || !c2w || (n && (!pix_i || !pix_j || !rays_o || !rays_d))) return ADFP_E_ARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_rays_from_uv, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pix_i, pix_j, n, fx, fy, cx, cy, c2w, rays_o, rays_d);
    ADFP_CHECK_LAUNCH();
    return 0;
}
int adfp_rays_from_uv_backward(const float* pix_i, const float* pix_j, int n, float fx, float fy, float cx, float cy, const float* g_rays_o,
                               const float* g_rays_d, float* g_c2w, void* stream) {
    if (n < 0 || !g_c2w || (n && (!pix_i || !pix_j))) return ADFP_E_ARG;
    hipLaunchKernelGGL(k_rays_from_uv_bwd, dim3(1), dim3(256), 0, (hipStream_t)stream, pix_i, pix_j, n, fx, fy, cx, cy, g_rays_o, g_rays_d, g_c2w);
    ADFP_CHECK_LAUNCH();
    return 0;
}

int adfp_prefilter_rays(const float* rays_o, const float* rays_d, const float* gt_depth, int n_rays, const double* bound_dev,
                        int* out_index, int* out_count, void* stream) {
    if (!out_count || n_rays < 0) return ADFP_E_ARG;
    if (n_rays == 0) return (int)zero_async(out_count, 4, (hipStream_t)stream);
    if (!rays_o || !rays_d || !gt_depth || !bound_dev || !out_index) return ADFP_E_ARG;
    hipLaunchKernelGGL(k_prefilter, dim3(1), dim3(1024), 0, (hipStream_t)stream, rays_o, rays_d, gt_depth, n_rays, bound_dev, out_index, out_count);
    ADFP_CHECK_LAUNCH();
    return 0;
}

static int sample_rays_impl(const float* rays_o, const float* rays_d, const float* gt_depth, int n_rays, const double bound[3][2],
                            int n_samples, int n_surface, int lindisp, float perturb, const float* t_rand, const float* depth_max,
                            double* z_vals, void* scratch, void* stream, bool scratch_is_zero, int segment = 0, int first_ray = 0,
                            const unsigned* seg_parts = nullptr, const RelayoutJobs* relayout = nullptr);
int adfp_sample_rays(const float* rays_o, const float* rays_d, const float* gt_depth, int n_rays, const double bound[3][2],
                     int n_samples, int n_surface, int lindisp, float perturb, const float* t_rand, const float* depth_max,
                     double* z_vals, void* scratch, void* stream) {
    return sample_rays_impl(rays_o, rays_d, gt_depth, n_rays, bound, n_samples, n_surface, lindisp, perturb, t_rand, depth_max, z_vals,
                            scratch, stream, false);
}
static int sample_rays_impl(const float* rays_o, const float* rays_d, const float* gt_depth, int n_rays, const double bound[3][2],
                            int n_samples, int n_surface, int lindisp, float perturb, const float* t_rand, const float* depth_max,
                            double* z_vals, void* scratch, void* stream, bool scratch_is_zero, int segment, int first_ray,
                            const unsigned* seg_parts, const RelayoutJobs* relayout) {
    if (!rays_o || !rays_d || !z_vals || !bound || n_rays < 0 || n_samples <= 0 || n_surface < 0) return ADFP_E_ARG;
    if (first_ray < 0 || (first_ray > 0 && ((!depth_max && !seg_parts) || segment <= 0))) return ADFP_E_ARG;
    if (perturb > 0.f && !t_rand) return ADFP_E_ARG;
    if (n_samples + n_surface > ADFP_MAX_SAMPLES) return ADFP_E_UNSUPPORTED;
    if (n_rays == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    SampleArgs a;
    a.ro = rays_o; a.rd = rays_d; a.depth = gt_depth; a.t_rand = perturb > 0.f ? t_rand : nullptr;
    a.dmax_f = depth_max; a.dmax_ord = nullptr; a.dmax_parts = nullptr; a.dmax_seg = segment > 0 ? segment : 0; a.dmax_first = first_ray;
    fill_bound(a.b, bound);
    a.n_rays = n_rays; a.n_samples = n_samples; a.n_surface = n_surface; a.lindisp = lindisp; a.perturb = perturb; a.z = z_vals;
    if (gt_depth && !depth_max && seg_parts) {          // the caller's first launch left SEGMAX_PARTS partial maxima per segment: k_sample folds them
        a.dmax_parts = seg_parts;
    } else if (gt_depth && !depth_max && segment > 0) {        // one maximum per segment, into the scratch (48 floats of room)
        if (!scratch) return ADFP_E_ARG;
        const int nseg = (n_rays + segment - 1) / segment;
        if (nseg > 48) return ADFP_E_UNSUPPORTED;
        if (!scratch_is_zero) { hipError_t e = zero_async(scratch, 192, st); if (e != hipSuccess) return (int)e; }
        int per = (segment + 2047) / 2048; if (per > 16) per = 16;     // few blocks per segment: the cost is the atomics' latency
        hipLaunchKernelGGL(k_depth_max_seg, dim3(per, nseg), dim3(256), 0, st, gt_depth, n_rays, segment, (unsigned*)scratch);
        ADFP_CHECK_LAUNCH();
        a.dmax_ord = (const unsigned*)scratch;
    } else if (gt_depth && !depth_max) {
        if (!scratch) return ADFP_E_ARG;
        if (!scratch_is_zero) { hipError_t e = zero_async(scratch, 16, st); if (e != hipSuccess) return (int)e; }
        int blocks = (n_rays + 2047) / 2048; if (blocks > 64) blocks = 64;     // few blocks: the cost is the atomics' latency
        hipLaunchKernelGGL(k_depth_max, dim3(blocks), dim3(256), 0, st, gt_depth, n_rays, (unsigned*)scratch);
        ADFP_CHECK_LAUNCH();
        a.dmax_ord = (const unsigned*)scratch;
    }
    a.nb_sample = (n_rays + 4 * ADFP_SAMPLE_RPW - 1) / (4 * ADFP_SAMPLE_RPW);
    a.rl.n = 0;
    unsigned nb_rl = 0;
    if (relayout && relayout->n > 0) { a.rl = *relayout; nb_rl = relayout->first[relayout->n]; }
    hipLaunchKernelGGL(k_sample, dim3((unsigned)a.nb_sample + nb_rl), dim3(256), 0, st, a);
    ADFP_CHECK_LAUNCH();
    return 0;
}

// corner-block copy of a TSDF volume (TriBlock, adfp_device.h): thread = voxel, z fastest like the reference's volume, so the
// eight strided reads of a wave are (mostly) the same lines its neighbours read and the 32-byte pieces it writes are consecutive
__global__ __launch_bounds__(256) void k_tsdf_corner_blocks(TsdfDev t, float* __restrict__ dst) {
    const long long n = (long long)t.X * t.Y * t.Z;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int z0 = (int)(idx % t.Z), y0 = (int)((idx / t.Z) % t.Y), x0 = (int)(idx / ((long long)t.Z * t.Y));
    const int x1 = x0 + 1 < t.X ? x0 + 1 : t.X - 1, y1 = y0 + 1 < t.Y ? y0 + 1 : t.Y - 1, z1 = z0 + 1 < t.Z ? z0 + 1 : t.Z - 1;
    const float* d = t.data;
    const long long ox0 = x0 * t.sX, ox1 = x1 * t.sX, oy0 = y0 * t.sY, oy1 = y1 * t.sY, oz0 = z0 * t.sZ, oz1 = z1 * t.sZ;
    const f32x4 lo = {d[oz0 + oy0 + ox0], d[oz0 + oy0 + ox1], d[oz0 + oy1 + ox0], d[oz0 + oy1 + ox1]};
    const f32x4 hi = {d[oz1 + oy0 + ox0], d[oz1 + oy0 + ox1], d[oz1 + oy1 + ox0], d[oz1 + oy1 + ox1]};
    f32x4* o = (f32x4*)(dst + idx * 8);
    o[0] = lo; o[1] = hi;
}
int adfp_relayout_tsdf(const adfp_tsdf* tsdf, float* dst, void* stream) {
    if (!tsdf || !tsdf->data || !dst || tsdf->X <= 0 || tsdf->Y <= 0 || tsdf->Z <= 0) return ADFP_E_ARG;
    const long long n = (long long)tsdf->X * tsdf->Y * tsdf->Z;
    if ((n + 255) / 256 > 0x7fffffffll) return ADFP_E_UNSUPPORTED;
    adfp_tsdf src = *tsdf; src.corner_blocks = nullptr;
    hipLaunchKernelGGL(k_tsdf_corner_blocks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, make_tsdf(src), dst);
    ADFP_CHECK_LAUNCH();
    return 0;
}

static int launch_tsdf(const adfp_scene* sc, const PtsDev& P, unsigned char* flags, int* list, float* att_u, float* w,
                       int* counter, float* tsdf_out, hipStream_t st) {
    TsdfArgs a;
    a.P = P; a.nt = make_norm(sc->tsdf_bnds); a.t = make_tsdf(sc->tsdf); fill_bound(a.b, sc->bound);
    a.flags = flags; a.list = list; a.att_u = att_u; a.w = w; a.counter = counter; a.tsdf_out = tsdf_out;
    if (P.n == 0) return 0;
    int blocks = (P.n + TSDF_CHUNK - 1) / TSDF_CHUNK;
    a.rays_per_block = 1;
    if (P.mode == ADFP_PTS_RAYS) {
        if (P.S > TSDF_CHUNK) return ADFP_E_UNSUPPORTED;
        int RB = 1;
        while (RB * 2 * P.S <= TSDF_CHUNK && RB < 64) RB *= 2;
        const int nr = P.n / P.S;
        // A block walks its points 256 at a time, so a small batch in few large blocks is a chain of dependent gathers on a handful
        // of CUs (200 rays x 64 samples in 7 blocks: 19 us; a Mapper batch of 1 000 rays: 11 us).  Fewer rays per block until there
        // are ~100 blocks -- not more: every block with in-band points ends in one atomic on the list counter, ~17 ns each in turn.
        while (RB > 1 && nr / RB < 96) RB >>= 1;
        a.rays_per_block = RB;
        blocks = (nr + RB - 1) / RB;
    }
    hipLaunchKernelGGL(k_tsdf, dim3(blocks), dim3(256), 0, st, a);
    ADFP_CHECK_LAUNCH();
    return 0;
}

int adfp_tsdf_stage(const adfp_scene* scene, const adfp_points* pts, unsigned char* flags, int* list, float* att_u, float* w,
                    int* counter, void* stream) {
    if (!scene || !pts || !scene->tsdf.data) return ADFP_E_ARG;
    if (list && (!att_u || !counter)) return ADFP_E_ARG;
    PtsDev P; int rc = make_pts(pts, &P); if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (counter) { hipError_t e = zero_async(counter, 4, st); if (e != hipSuccess) return (int)e; }
    return launch_tsdf(scene, P, flags, list, att_u, w, counter, nullptr, st);
}

int adfp_sample_tsdf(const adfp_tsdf* tsdf, const double tsdf_bnds[3][2], const adfp_points* pts, float* out, void* stream) {
    if (!tsdf || !tsdf->data || !tsdf_bnds || !pts || !out) return ADFP_E_ARG;
    PtsDev P; int rc = make_pts(pts, &P); if (rc) return rc;
    adfp_scene sc; memset(&sc, 0, sizeof(sc));
    sc.tsdf = *tsdf;
    for (int k = 0; k < 3; ++k) { sc.tsdf_bnds[k][0] = tsdf_bnds[k][0]; sc.tsdf_bnds[k][1] = tsdf_bnds[k][1]; sc.bound[k][0] = 0; sc.bound[k][1] = 1; }
    return launch_tsdf(&sc, P, nullptr, nullptr, nullptr, nullptr, nullptr, out, (hipStream_t)stream);
}

#ifdef ADFP_STAMPS
extern "C" int adfp_debug_stamps(unsigned long long* host_out, int n_waves) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), (size_t)n_waves * 16);
}
extern "C" int adfp_debug_phases(unsigned long long* host_out, int reset) {
    int rc = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase), 64);
    if (!rc && reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, 64); }
    return rc;
}
#endif

static int decode_grid(int ntiles, int waves_per_wg, int wg_per_cu) {
    int g = (ntiles + waves_per_wg - 1) / waves_per_wg;
    const int cap = num_cu() * wg_per_cu;
    if (g > cap) g = cap;
    return g < 1 ? 1 : g;
}

static int eval_points_impl(const adfp_scene* sc, const PtsDev& P, int stage, int apply_bound, float* raw, float* w, Workspace& ws_in, hipStream_t st,
                            const adfp_train_state* state = nullptr, bool ws_counter_is_zero = false) {
    if (P.n == 0) return 0;
    Workspace ws = ws_in;
    if (state) {       // training: the backward needs these buffers after the call returns
        ws.flags = state->flags; ws.list = state->list; ws.counter = state->counter;
        ws.att_occ = state->att_occ; ws.att_u = state->att_u;
    }
    hipError_t e;
    const bool fuse = stage != ADFP_STAGE_LOW;
    if (stage != ADFP_STAGE_COLOR) {           // rgb = 0 in stages low/high (decoder.py:317, :323)
        e = zero_async(raw, (size_t)P.n * 16, st);
        if (e != hipSuccess) return (int)e;
    }
    // f16-split kernels in this call?  Then the call owns a range flag (counter[8]) that they raise, k_fallback_points reads, and --
    // in a training call, where `counter` is the caller's -- the backward entries and the Adam step gate on.
    const bool any_h = sc->h_low || (fuse && (sc->h_high || sc->h_att)) || (stage == ADFP_STAGE_COLOR && sc->h_color);
    int* call_flag = (any_h && ws.counter) ? ws.counter + 8 : nullptr;
    if ((fuse || any_h || state) && ws.counter && !ws_counter_is_zero) {      // (a training call's counter is the caller's buffer: ws.counter above)
        e = zero_async(ws.counter, 64, st);
        if (e != hipSuccess) return (int)e;
    }
    if (fuse) {
        int rc = launch_tsdf(sc, P, ws.flags, ws.list, ws.att_u, nullptr, ws.counter, nullptr, st);   // w = 1 comes from the LOW decoder
        if (rc) return rc;
    }
    DecodeArgs a;
    a.P = P; a.nb = make_norm(sc->bound); fill_bound(a.b, sc->bound);
    a.list = nullptr; a.count_ptr = nullptr; a.flags = fuse ? ws.flags : nullptr;
    a.raw = raw; a.w = w; a.att_occ = nullptr; a.write_w = 1; a.apply_bound = apply_bound;   // attention overwrites w on the band
    a.status = sc->status; a.masks = nullptr; a.act = nullptr; a.single = 0; a.call_flag = call_flag;
    const int ntiles = (P.n + 31) / 32;
    // stage color, inference, both networks f16-split: LOW and COLOR on every point in ONE launch (k_decode_lc)
    const bool fused_lc = stage == ADFP_STAGE_COLOR && !state && sc->h_low && sc->h_color;
    if (fused_lc) {
        DecodeLCArgs f;
        f.P = P; f.nb = a.nb; fill_bound(f.b, sc->bound);
        f.g_low = make_grid(sc->low); f.g_color = make_grid(sc->color);
        f.packed_low = (const unsigned*)sc->h_low; f.packed_color = (const unsigned*)sc->h_color;
        f.flags = a.flags; f.raw = raw; f.w = w; f.write_w = 1; f.apply_bound = apply_bound; f.status = sc->status; f.call_flag = call_flag;
        f.pool = ws.counter ? ws.counter + 10 : nullptr;            // zero: this call's 64-byte counter block was cleared above (or by the caller)
#ifdef ADFP_LC_32X32          // A/B build: the 32x32x16 form of the fused launch (adfp_decode_h.h)
        hipLaunchKernelGGL((k_decode_lc<ADFP_LC_NT>), dim3(decode_grid(ntiles, ADFP_LC_NT / 64, 1)), dim3(ADFP_LC_NT), 0, st, f);
#else
        f.packed_low += DecLayoutH<32, 1>::P_TOTAL; f.packed_color += DecLayoutH<32, 4>::P_TOTAL;       // the G images
        hipLaunchKernelGGL((k_decode_lc16<ADFP_LC_NT>), dim3(decode_grid(ntiles, ADFP_LC_NT / 64, 1)), dim3(ADFP_LC_NT), 0, st, f);
#endif
        ADFP_CHECK_LAUNCH();
    }
    // stage color, TRAINING, both networks f16-split with mask room: the same ONE launch, leaving masks (+ layer inputs)
#ifdef ADFP_LC_32X32
    const bool fused_lc_train = false;
#else
    const bool fused_lc_train = stage == ADFP_STAGE_COLOR && state && sc->h_low && sc->h_color && state->masks_low && state->masks_color;
    if (fused_lc_train) {
        DecodeLCTrainArgs t;
        DecodeLCArgs& f = t.f;
        f.P = P; f.nb = a.nb; fill_bound(f.b, sc->bound);
        f.g_low = make_grid(sc->low); f.g_color = make_grid(sc->color);
        f.packed_low = (const unsigned*)sc->h_low + DecLayoutH<32, 1>::P_TOTAL; f.packed_color = (const unsigned*)sc->h_color + DecLayoutH<32, 4>::P_TOTAL;   // the G images
        f.flags = a.flags; f.raw = raw; f.w = w; f.write_w = 1; f.apply_bound = apply_bound; f.status = sc->status; f.call_flag = call_flag; f.pool = nullptr;
        t.masks_low = state->masks_low; t.masks_color = state->masks_color; t.act_low = state->act_low; t.act_color = state->act_color;
        // few tiles (every (tile, network) pair can have a wave of its own): one network per wave
        t.split_networks = 2 * ntiles <= num_cu() * (ADFP_LCT_NT / 64) ? 1 : 0;
        if (t.split_networks) hipLaunchKernelGGL((k_decode_lc16_train<ADFP_LCT_NT, true>), dim3(decode_grid(2 * ntiles, ADFP_LCT_NT / 64, 1)), dim3(ADFP_LCT_NT), 0, st, t);
        else hipLaunchKernelGGL((k_decode_lc16_train<ADFP_LCT_NT, false>), dim3(decode_grid(ntiles, ADFP_LCT_NT / 64, 1)), dim3(ADFP_LCT_NT), 0, st, t);
        ADFP_CHECK_LAUNCH();
    }
#endif
    // LOW on every point
    a.g0 = make_grid(sc->low); a.g1 = a.g0;
    if (fused_lc || fused_lc_train) {
    } else if (sc->h_low && state && state->masks_low) {              // training forward: leaves the ReLU masks (+ layer inputs)
        a.packed = (const float*)sc->h_low; a.masks = state->masks_low; a.act = state->act_low;
        if (a.act) hipLaunchKernelGGL((k_decode_h<32, 1, ROLE_LOW, 512, 2>), dim3(decode_grid(ntiles, 8, 1)), dim3(512), 0, st, a);
        else hipLaunchKernelGGL((k_decode_h<32, 1, ROLE_LOW, 512, 1>), dim3(decode_grid(ntiles, 8, 1)), dim3(512), 0, st, a);
        a.masks = nullptr; a.act = nullptr;
    } else if (sc->h_low) {
        a.packed = (const float*)sc->h_low;
            hipLaunchKernelGGL((k_decode_h<32, 1, ROLE_LOW, ADFP_DECH_NT>), dim3(decode_grid(ntiles, ADFP_DECH_NT / 64, ADFP_DECH_WG)), dim3(ADFP_DECH_NT), 0, st, a);
    } else {
        a.packed = sc->w_low;
        hipLaunchKernelGGL((k_decode<32, 1, ROLE_LOW, 256>), dim3(decode_grid(ntiles, 4, 2)), dim3(256), 0, st, a);
    }
    ADFP_CHECK_LAUNCH();
    if (stage == ADFP_STAGE_COLOR && !fused_lc && !fused_lc_train) {
        a.g0 = make_grid(sc->color); a.g1 = a.g0;
        if (sc->h_color && state && state->masks_color) {
            a.packed = (const float*)sc->h_color; a.masks = state->masks_color; a.act = state->act_color;
            if (a.act) hipLaunchKernelGGL((k_decode_h<32, 4, ROLE_COLOR, 512, 2>), dim3(decode_grid(ntiles, 8, 1)), dim3(512), 0, st, a);
            else hipLaunchKernelGGL((k_decode_h<32, 4, ROLE_COLOR, 512, 1>), dim3(decode_grid(ntiles, 8, 1)), dim3(512), 0, st, a);
            a.masks = nullptr; a.act = nullptr;
        } else if (sc->h_color) {
            a.packed = (const float*)sc->h_color;
            hipLaunchKernelGGL((k_decode_h<32, 4, ROLE_COLOR, ADFP_DECH_NT>), dim3(decode_grid(ntiles, ADFP_DECH_NT / 64, ADFP_DECH_WG)), dim3(ADFP_DECH_NT), 0, st, a);
        } else {
            a.packed = sc->w_color;
            hipLaunchKernelGGL((k_decode<32, 4, ROLE_COLOR, 256>), dim3(decode_grid(ntiles, 4, 2)), dim3(256), 0, st, a);
        }
        ADFP_CHECK_LAUNCH();
    }
    if (fuse) {
        a.g0 = make_grid(sc->high); a.g1 = make_grid(sc->low);
        a.list = ws.list; a.count_ptr = ws.counter; a.att_occ = ws.att_occ;
        if (sc->h_high && state && state->masks_high) {
            a.packed = (const float*)sc->h_high; a.masks = state->masks_high; a.act = state->act_high;
            if (a.act) hipLaunchKernelGGL((k_decode_h<64, 1, ROLE_HIGH, 512, 2>), dim3(decode_grid(ntiles, 8, 1)), dim3(512), 0, st, a);
            else hipLaunchKernelGGL((k_decode_h<64, 1, ROLE_HIGH, 512, 1>), dim3(decode_grid(ntiles, 8, 1)), dim3(512), 0, st, a);
            a.masks = nullptr; a.act = nullptr;
        } else if (sc->h_high) {
#ifdef ADFP_LC_32X32
            a.packed = (const float*)sc->h_high;
            hipLaunchKernelGGL((k_decode_h<64, 1, ROLE_HIGH, ADFP_HIGH_NT>), dim3(decode_grid(ntiles, ADFP_HIGH_NT / 64, 1)), dim3(ADFP_HIGH_NT), 0, st, a);
#else
            a.packed = (const float*)((const unsigned*)sc->h_high + DecLayoutH<64, 1>::P_TOTAL);          // the G image
#ifdef ADFP_EXP_HIGH_AS_LOW
            a.packed = (const float*)((const unsigned*)sc->h_low + DecLayoutH<32, 1>::P_TOTAL);           // timing experiment: see k_decode_high_g
#endif
            a.pool = ws.counter ? ws.counter + 11 : nullptr;
            hipLaunchKernelGGL((k_decode_high_g<ADFP_HIGH_NT>), dim3(decode_grid(ntiles, ADFP_HIGH_NT / 64, 1)), dim3(ADFP_HIGH_NT), 0, st, a);
#endif
        } else {
            a.packed = sc->w_high;
            hipLaunchKernelGGL((k_decode<64, 1, ROLE_HIGH, 512>), dim3(decode_grid(ntiles, 8, 1)), dim3(512), 0, st, a);
        }
        ADFP_CHECK_LAUNCH();
        AttArgs t;
        t.list = ws.list; t.count_ptr = ws.counter; t.att_occ = ws.att_occ; t.att_u = ws.att_u;
        t.flags = ws.flags; t.raw = raw; t.w = w; t.apply_bound = apply_bound; t.status = sc->status; t.n_rows = 0; t.call_flag = call_flag;
        t.masks = nullptr; t.act = nullptr;
        if (sc->h_att && state && state->masks_att) {
            t.packed = (const float*)sc->h_att; t.masks = state->masks_att; t.act = state->act_att;
            hipLaunchKernelGGL((k_attention_h<1, 256>), dim3(decode_grid(ntiles, 4, 1)), dim3(256), 0, st, t);
        } else if (sc->h_att) {
#ifdef ADFP_LC_32X32
            t.packed = (const float*)sc->h_att;
            hipLaunchKernelGGL(k_attention_h<0>, dim3(decode_grid(ntiles, 8, 1)), dim3(512), 0, st, t);
#else
            t.packed = (const float*)((const unsigned*)sc->h_att + AttLayoutH::P_TOTAL);                 // the G image
            t.pool = ws.counter ? ws.counter + 12 : nullptr;
            hipLaunchKernelGGL(k_attention_g<512>, dim3(decode_grid(ntiles, 8, 1)), dim3(512), 0, st, t);
#endif
        } else {
            t.packed = sc->w_att;
            hipLaunchKernelGGL(k_attention, dim3(decode_grid(ntiles, 8, 1)), dim3(512), 0, st, t);
        }
        ADFP_CHECK_LAUNCH();
    }
    // the f32 repair of a call whose f16-split kernels left the f16 range (adfp_fallback.h): returns at once when the flag is clear
    if (call_flag && sc->flat_low && (!fuse || (sc->flat_high && sc->flat_att)) && (stage != ADFP_STAGE_COLOR || sc->flat_color)) {
        FallbackArgs f;
        f.P = P; f.nb = a.nb; fill_bound(f.b, sc->bound);
        f.low = make_grid(sc->low); f.high = make_grid(sc->high); f.color = make_grid(sc->color);
        f.flat_low = sc->flat_low; f.flat_high = sc->flat_high; f.flat_color = sc->flat_color; f.flat_att = sc->flat_att;
        f.stage = stage; f.apply_bound = apply_bound;
        f.flags = fuse ? ws.flags : nullptr; f.list = ws.list; f.count_ptr = ws.counter; f.att_u = ws.att_u; f.att_occ = ws.att_occ;
        f.raw = raw; f.w = w; f.call_flag = call_flag;
        long long blocks = ((long long)P.n * (fuse ? 2 : 1) + 255) / 256;
        const long long cap = (long long)num_cu() * 8;
        hipLaunchKernelGGL(k_fallback_points, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(256), 0, st, f);
        ADFP_CHECK_LAUNCH();
    }
    return 0;
}

// bench hook: ONE decoder kernel (LOW or COLOR) over every point, nothing else.
extern "C" int adfp_decode_stage(const adfp_scene* sc, const adfp_points* pts, int kind, float* raw, float* w, int* tile_counter, void* stream) {
    if (!sc || !pts || !raw || !w) return ADFP_E_ARG;
    if (kind != ADFP_DEC_LOW && kind != ADFP_DEC_COLOR && kind != ADFP_DEC_LOW_COLOR) return ADFP_E_UNSUPPORTED;
    if (grid_too_big(sc->low) || grid_too_big(sc->color)) return ADFP_E_UNSUPPORTED;
    PtsDev P; int rc = make_pts(pts, &P); if (rc) return rc;
    if (P.n == 0) return 0;
    if (kind == ADFP_DEC_LOW_COLOR) {                  // the fused launch of stage color (f16-split images only)
        if (!sc->low.data || !sc->color.data || !sc->h_low || !sc->h_color) return ADFP_E_ARG;
        DecodeLCArgs f;
        f.P = P; f.nb = make_norm(sc->bound); fill_bound(f.b, sc->bound);
        f.g_low = make_grid(sc->low); f.g_color = make_grid(sc->color);
        f.packed_low = (const unsigned*)sc->h_low; f.packed_color = (const unsigned*)sc->h_color;
        f.flags = nullptr; f.raw = raw; f.w = w; f.write_w = 1; f.apply_bound = 1; f.status = sc->status; f.call_flag = nullptr; f.pool = tile_counter;
        if (tile_counter) { hipError_t e = zero_async(tile_counter, 4, (hipStream_t)stream); if (e != hipSuccess) return (int)e; }
#ifdef ADFP_LC_32X32
        hipLaunchKernelGGL((k_decode_lc<ADFP_LC_NT>), dim3(decode_grid((P.n + 31) / 32, ADFP_LC_NT / 64, 1)), dim3(ADFP_LC_NT), 0, (hipStream_t)stream, f);
#else
        f.packed_low += DecLayoutH<32, 1>::P_TOTAL; f.packed_color += DecLayoutH<32, 4>::P_TOTAL;       // the G images
        hipLaunchKernelGGL((k_decode_lc16<ADFP_LC_NT>), dim3(decode_grid((P.n + 31) / 32, ADFP_LC_NT / 64, 1)), dim3(ADFP_LC_NT), 0, (hipStream_t)stream, f);
#endif
        ADFP_CHECK_LAUNCH();
        return 0;
    }
    DecodeArgs a;
    a.P = P; a.nb = make_norm(sc->bound); fill_bound(a.b, sc->bound);
    a.list = nullptr; a.count_ptr = nullptr; a.flags = nullptr;
    a.raw = raw; a.w = w; a.att_occ = nullptr; a.write_w = 1; a.apply_bound = 1; a.status = sc->status;
    a.masks = nullptr; a.act = nullptr; a.single = 0; a.call_flag = nullptr;
    const int ntiles = (P.n + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    if (kind == ADFP_DEC_LOW) {
        if (!sc->low.data || !(sc->w_low || sc->h_low)) return ADFP_E_ARG;
        a.g0 = make_grid(sc->low); a.g1 = a.g0;
        if (sc->h_low) {
            a.packed = (const float*)sc->h_low;
            hipLaunchKernelGGL((k_decode_h<32, 1, ROLE_LOW, ADFP_DECH_NT>), dim3(decode_grid(ntiles, ADFP_DECH_NT / 64, ADFP_DECH_WG)), dim3(ADFP_DECH_NT), 0, st, a);
        } else {
            a.packed = sc->w_low;
            hipLaunchKernelGGL((k_decode<32, 1, ROLE_LOW, 256>), dim3(decode_grid(ntiles, 4, 2)), dim3(256), 0, st, a);
        }
    } else {
        if (!sc->color.data || !(sc->w_color || sc->h_color)) return ADFP_E_ARG;
        a.g0 = make_grid(sc->color); a.g1 = a.g0;
        if (sc->h_color) {
            a.packed = (const float*)sc->h_color;
            hipLaunchKernelGGL((k_decode_h<32, 4, ROLE_COLOR, ADFP_DECH_NT>), dim3(decode_grid(ntiles, ADFP_DECH_NT / 64, ADFP_DECH_WG)), dim3(ADFP_DECH_NT), 0, st, a);
        } else {
            a.packed = sc->w_color;
            hipLaunchKernelGGL((k_decode<32, 4, ROLE_COLOR, 256>), dim3(decode_grid(ntiles, 4, 2)), dim3(256), 0, st, a);
        }
    }
    ADFP_CHECK_LAUNCH();
    return 0;
}

// ---- one sub-network alone: what `decoders.low_decoder(p, c_grid)` / `decoders.mlp(p, occ, tsdf_volume, tsdf_bnds)` compute in
// the reference (decoder.py:177-203, :240-258).  No bound rule, no band logic.
__global__ __launch_bounds__(256) void k_inv_tsdf(const float* __restrict__ t, float* __restrict__ u, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) u[i] = inv_tsdf(t[i]);
}
extern "C" int adfp_decode_single(const adfp_scene* sc, const adfp_points* pts, int kind, float* out4, void* stream) {
    if (!sc || !pts || !out4) return ADFP_E_ARG;
    if (grid_too_big(sc->low) || grid_too_big(sc->high) || grid_too_big(sc->color)) return ADFP_E_UNSUPPORTED;
    PtsDev P; int rc = make_pts(pts, &P); if (rc) return rc;
    if (P.n == 0) return 0;
    DecodeArgs a;
    a.P = P; a.nb = make_norm(sc->bound); fill_bound(a.b, sc->bound);
    a.list = nullptr; a.count_ptr = nullptr; a.flags = nullptr;
    a.raw = out4; a.w = nullptr; a.att_occ = nullptr; a.write_w = 0; a.apply_bound = 0; a.status = sc->status;
    a.masks = nullptr; a.act = nullptr; a.single = 1; a.call_flag = nullptr;
    const int ntiles = (P.n + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    if (kind == ADFP_DEC_LOW) {
        if (!sc->low.data || !(sc->w_low || sc->h_low)) return ADFP_E_ARG;
        a.g0 = make_grid(sc->low); a.g1 = a.g0;
        if (sc->h_low) { a.packed = (const float*)sc->h_low; hipLaunchKernelGGL((k_decode_h<32, 1, ROLE_LOW, ADFP_DECH_NT>), dim3(decode_grid(ntiles, ADFP_DECH_NT / 64, ADFP_DECH_WG)), dim3(ADFP_DECH_NT), 0, st, a); }
        else { a.packed = sc->w_low; hipLaunchKernelGGL((k_decode<32, 1, ROLE_LOW, 256>), dim3(decode_grid(ntiles, 4, 2)), dim3(256), 0, st, a); }
    } else if (kind == ADFP_DEC_COLOR) {
        if (!sc->color.data || !(sc->w_color || sc->h_color)) return ADFP_E_ARG;
        a.g0 = make_grid(sc->color); a.g1 = a.g0;
        if (sc->h_color) { a.packed = (const float*)sc->h_color; hipLaunchKernelGGL((k_decode_h<32, 4, ROLE_COLOR, ADFP_DECH_NT>), dim3(decode_grid(ntiles, ADFP_DECH_NT / 64, ADFP_DECH_WG)), dim3(ADFP_DECH_NT), 0, st, a); }
        else { a.packed = sc->w_color; hipLaunchKernelGGL((k_decode<32, 4, ROLE_COLOR, 256>), dim3(decode_grid(ntiles, 4, 2)), dim3(256), 0, st, a); }
    } else if (kind == ADFP_DEC_HIGH) {
        if (!sc->high.data || !sc->low.data || !(sc->w_high || sc->h_high)) return ADFP_E_ARG;
        a.g0 = make_grid(sc->high); a.g1 = make_grid(sc->low);
        a.att_occ = out4;                                   // HIGH writes one float per point: out4 is [P] here
        a.raw = nullptr;
        if (sc->h_high) { a.packed = (const float*)sc->h_high; hipLaunchKernelGGL((k_decode_h<64, 1, ROLE_HIGH, 512>), dim3(decode_grid(ntiles, 8, 1)), dim3(512), 0, st, a); }
        else { a.packed = sc->w_high; hipLaunchKernelGGL((k_decode<64, 1, ROLE_HIGH, 512>), dim3(decode_grid(ntiles, 8, 1)), dim3(512), 0, st, a); }
    } else return ADFP_E_ARG;
    ADFP_CHECK_LAUNCH();
    return 0;
}
extern "C" int adfp_attention_rows(const adfp_scene* sc, const float* occ, const float* tsdf_val, long long n, float* out4, float* w,
                                   float* scratch_u, void* stream) {
    if (!sc || n < 0 || (n && (!occ || !tsdf_val || !out4 || !w || !scratch_u)) || !(sc->w_att || sc->h_att)) return ADFP_E_ARG;
    if (n > 0x7fffffffll) return ADFP_E_UNSUPPORTED;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_inv_tsdf, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tsdf_val, scratch_u, n);
    ADFP_CHECK_LAUNCH();
    AttArgs t;
    t.list = nullptr; t.count_ptr = nullptr; t.att_occ = occ; t.att_u = scratch_u; t.flags = nullptr;
    t.raw = out4; t.w = w; t.apply_bound = 0; t.n_rows = (int)n; t.status = sc->status; t.masks = nullptr; t.act = nullptr; t.call_flag = nullptr;
    const int ntiles = (int)((n + 31) / 32);
    if (sc->h_att) { t.packed = (const float*)sc->h_att; hipLaunchKernelGGL(k_attention_h<0>, dim3(decode_grid(ntiles, 8, 1)), dim3(512), 0, st, t); }
    else { t.packed = sc->w_att; hipLaunchKernelGGL(k_attention, dim3(decode_grid(ntiles, 8, 1)), dim3(512), 0, st, t); }
    ADFP_CHECK_LAUNCH();
    return 0;
}

static int check_scene(const adfp_scene* sc, int stage) {
    if (!sc) return ADFP_E_ARG;
    if (stage < ADFP_STAGE_LOW || stage > ADFP_STAGE_COLOR) return ADFP_E_ARG;
    if (grid_too_big(sc->low) || grid_too_big(sc->high) || grid_too_big(sc->color)) return ADFP_E_UNSUPPORTED;
    if (!sc->low.data || !(sc->w_low || sc->h_low)) return ADFP_E_ARG;
    if (stage >= ADFP_STAGE_HIGH && (!sc->high.data || !(sc->w_high || sc->h_high) || !(sc->w_att || sc->h_att) || !sc->tsdf.data)) return ADFP_E_ARG;
    if (stage == ADFP_STAGE_COLOR && (!sc->color.data || !(sc->w_color || sc->h_color))) return ADFP_E_ARG;
    return 0;
}

int adfp_eval_points_train(const adfp_scene* scene, const adfp_points* pts, int stage, int flags, float* raw, float* w,
                           void* workspace, size_t workspace_bytes, const adfp_train_state* state, void* stream) {
    int rc = check_scene(scene, stage); if (rc) return rc;
    if (!raw || !w || !workspace) return ADFP_E_ARG;
    if (state && stage != ADFP_STAGE_LOW && (!state->flags || !state->list || !state->counter || !state->att_occ || !state->att_u)) return ADFP_E_ARG;
    PtsDev P; rc = make_pts(pts, &P); if (rc) return rc;
    Workspace ws = carve(workspace, P.n);
    if (workspace_bytes < ws.bytes) return ADFP_E_WORKSPACE;
    return eval_points_impl(scene, P, stage, (flags & ADFP_EVAL_APPLY_BOUND) ? 1 : 0, raw, w, ws, (hipStream_t)stream, state);
}
int adfp_eval_points(const adfp_scene* scene, const adfp_points* pts, int stage, int flags, float* raw, float* w,
                     void* workspace, size_t workspace_bytes, void* stream) {
    return adfp_eval_points_train(scene, pts, stage, flags, raw, w, workspace, workspace_bytes, nullptr, stream);
}

int adfp_tsdf_integrate(float* tsdf, float* weight, float* color, int dim_x, int dim_y, int dim_z, const float origin[3], float voxel_size,
                        const float cam_intr[9], const float cam_pose[16], const float* color_im, const float* depth_im, int im_h, int im_w,
                        float trunc_margin, float obs_weight, void* stream) {
    if (!tsdf || !weight || !origin || !cam_intr || !cam_pose || !depth_im || (color && !color_im)) return ADFP_E_ARG;
    if (dim_x <= 0 || dim_y <= 0 || dim_z <= 0 || im_h <= 0 || im_w <= 0 || !(voxel_size > 0) || !(trunc_margin > 0)) return ADFP_E_ARG;
    const long long n = (long long)dim_x * dim_y * dim_z;
    if (n > 0x7fffffffll) return ADFP_E_UNSUPPORTED;
    FuseFrame f;
    f.sdf = tsdf; f.wsum = weight; f.rgb = color; f.nx = dim_x; f.ny = dim_y; f.nz = dim_z;
    for (int k = 0; k < 3; ++k) f.org[k] = origin[k];
    f.cell = voxel_size;
    for (int k = 0; k < 9; ++k) f.K[k] = cam_intr[k];
    for (int k = 0; k < 16; ++k) f.T[k] = cam_pose[k];
    f.rgb_im = color_im; f.z_im = depth_im; f.rows = im_h; f.cols = im_w; f.band = trunc_margin; f.w_obs = obs_weight;
    const long long quads = (n + 3) / 4;                     // a lane owns four z-consecutive voxels (adfp_fusion.h)
    const dim3 grid((unsigned)((quads + 255) / 256));
    const bool aligned = ((((uintptr_t)tsdf) | ((uintptr_t)weight) | ((uintptr_t)color)) & 15u) == 0;
    if (aligned) hipLaunchKernelGGL(k_fuse_frame<true>, grid, dim3(256), 0, (hipStream_t)stream, f);
    else hipLaunchKernelGGL(k_fuse_frame<false>, grid, dim3(256), 0, (hipStream_t)stream, f);
    ADFP_CHECK_LAUNCH();
    return 0;
}

int adfp_frustum_mask(int X, int Y, int Z, const double bound[3][2], const float c2w[16], const float w2c[16], double fx, double fy,
                      double cx, double cy, int H, int W, const float* depth, float* sampled, unsigned* scratch, unsigned char* mask,
                      void* stream) {
    if (!bound || !c2w || !w2c || !depth || !sampled || !scratch || !mask || X <= 0 || Y <= 0 || Z <= 0 || H <= 0 || W <= 0) return ADFP_E_ARG;
    FrustumArgs a;
    a.X = X; a.Y = Y; a.Z = Z;
    for (int k = 0; k < 3; ++k) { a.lo[k] = bound[k][0]; a.hi[k] = bound[k][1]; a.cam[k] = c2w[4 * k + 3]; }
    for (int k = 0; k < 16; ++k) a.w2c[k] = w2c[k];
    a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy; a.H = H; a.W = W; a.depth = depth;
    a.sampled = sampled; a.dmax_ord = scratch; a.mask = mask;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = zero_async(scratch, 4, st);
    if (e != hipSuccess) return (int)e;
    const long long n = (long long)X * Y * Z;
    long long blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_frustum_depth, dim3((unsigned)blocks), dim3(256), 0, st, a);
    ADFP_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_frustum_mask, dim3((unsigned)blocks), dim3(256), 0, st, a);
    ADFP_CHECK_LAUNCH();
    return 0;
}

int adfp_masked_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const unsigned char* mask, long long nvox,
                     int channels, float lr, float beta1, float beta2, float eps, int step, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || nvox < 0 || channels <= 0 || step < 1) return ADFP_E_ARG;
    if (nvox == 0) return 0;
    AdamArgs a;
    a.param = param; a.grad = grad; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq; a.mask = mask; a.nvox = nvox; a.C = channels;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.derived = nullptr;
    // bias corrections in double like torch.optim (python floats), then one rounding to f32
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.step_size = (float)((double)lr / bc1);
    a.sqrt_bc2 = (float)sqrt(bc2);
    const long long threads = ((nvox + 3) >> 2) * channels;
    hipLaunchKernelGGL(k_masked_adam, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    ADFP_CHECK_LAUNCH();
    return 0;
}

int adfp_adam_prep(int* steps, float* derived, int n_groups, const float* lr, float beta1, float beta2, const int* skip_flag, void* stream) {
    if (!steps || !derived || !lr || n_groups <= 0 || n_groups > ADFP_ADAM_MAX_GROUPS) return ADFP_E_ARG;
    AdamPrepArgs a;
    a.steps = steps; a.derived = derived; a.n = n_groups; a.beta1 = beta1; a.beta2 = beta2; a.skip = skip_flag;
    for (int g = 0; g < ADFP_ADAM_MAX_GROUPS; ++g) a.lr[g] = g < n_groups ? lr[g] : -1.f;
    hipLaunchKernelGGL(k_adam_prep, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    ADFP_CHECK_LAUNCH();
    return 0;
}
int adfp_masked_adam_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const unsigned char* mask, long long nvox,
                         int channels, float beta1, float beta2, float eps, const float* derived, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !derived || nvox < 0 || channels <= 0) return ADFP_E_ARG;
    if (nvox == 0) return 0;
    AdamArgs a;
    a.param = param; a.grad = grad; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq; a.mask = mask; a.nvox = nvox; a.C = channels;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.step_size = 0.f; a.sqrt_bc2 = 1.f; a.derived = derived;
    const long long threads = ((nvox + 3) >> 2) * channels;
    hipLaunchKernelGGL(k_masked_adam, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    ADFP_CHECK_LAUNCH();
    return 0;
}
static int adam_multi_table(int n_groups, const adfp_adam_group* groups, float beta1, float beta2, float eps, AdamMultiArgs& m) {
    if (n_groups < 0 || n_groups > ADFP_ADAM_MULTI || (n_groups && !groups)) return ADFP_E_ARG;
    m.n = 0; m.first_block[0] = 0;
    for (int k = 0; k < n_groups; ++k) {
        const adfp_adam_group& g = groups[k];
        if (!g.param || !g.grad || !g.exp_avg || !g.exp_avg_sq || !g.derived || g.nvox < 0 || g.channels <= 0) return ADFP_E_ARG;
        if (g.nvox == 0) continue;
        AdamArgs& a = m.g[m.n];
        a.param = g.param; a.grad = g.grad; a.exp_avg = g.exp_avg; a.exp_avg_sq = g.exp_avg_sq; a.mask = g.mask; a.nvox = g.nvox; a.C = g.channels;
        a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.step_size = 0.f; a.sqrt_bc2 = 1.f; a.derived = g.derived;
        const long long threads = ((g.nvox + 3) >> 2) * g.channels;
        m.first_block[m.n + 1] = m.first_block[m.n] + (unsigned)((threads + 255) / 256);
        ++m.n;
    }
    return 0;
}
static int adam_cl_table(int n_groups, const adfp_adam_cl_group* groups, float beta1, float beta2, float eps, AdamClMultiArgs& m) {
    if (n_groups < 0 || n_groups > ADFP_ADAM_CL_MULTI || (n_groups && !groups)) return ADFP_E_ARG;
    m.n = 0; m.first_block[0] = 0;
    for (int k = 0; k < n_groups; ++k) {
        const adfp_adam_cl_group& g = groups[k];
        if (!g.param_cl || !g.param_cm || !g.grad_cl || !g.exp_avg_cl || !g.exp_avg_sq_cl || !g.derived || g.nvox < 0) return ADFP_E_ARG;
        if (g.nvox == 0) continue;
        AdamClArgs& a = m.g[m.n];
        a.p_cl = g.param_cl; a.p_cm = g.param_cm; a.g_cl = g.grad_cl; a.m_cl = g.exp_avg_cl; a.v_cl = g.exp_avg_sq_cl; a.mask = g.mask;
        a.nvox = g.nvox; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.derived = g.derived;
        m.first_block[m.n + 1] = m.first_block[m.n] + (unsigned)((g.nvox + 63) / 64);
        ++m.n;
    }
    return 0;
}
int adfp_masked_adam_multi(int n_groups, const adfp_adam_group* groups, float beta1, float beta2, float eps, void* stream) {
    AdamMultiArgs m;
    int rc = adam_multi_table(n_groups, groups, beta1, beta2, eps, m); if (rc) return rc;
    if (m.n == 0) return 0;
    hipLaunchKernelGGL(k_masked_adam_multi, dim3(m.first_block[m.n]), dim3(256), 0, (hipStream_t)stream, m);
    ADFP_CHECK_LAUNCH();
    return 0;
}
int adfp_adam_grids_cl(int n_groups, const adfp_adam_cl_group* groups, float beta1, float beta2, float eps, void* stream) {
    AdamClMultiArgs m;
    int rc = adam_cl_table(n_groups, groups, beta1, beta2, eps, m); if (rc) return rc;
    if (m.n == 0) return 0;
    hipLaunchKernelGGL(k_adam_cl_multi, dim3(m.first_block[m.n]), dim3(256), 0, (hipStream_t)stream, m);
    ADFP_CHECK_LAUNCH();
    return 0;
}
int adfp_adam_step(int n_cl_groups, const adfp_adam_cl_group* cl_groups, int n_groups, const adfp_adam_group* groups, float beta1, float beta2,
                   float eps, void* stream) {
    AdamStepArgs s;
    int rc = adam_cl_table(n_cl_groups, cl_groups, beta1, beta2, eps, s.cl); if (rc) return rc;
    rc = adam_multi_table(n_groups, groups, beta1, beta2, eps, s.fl); if (rc) return rc;
    s.cl_blocks = s.cl.n ? s.cl.first_block[s.cl.n] : 0u;
    const unsigned fl_blocks = s.fl.n ? s.fl.first_block[s.fl.n] : 0u;
    if (s.cl_blocks + fl_blocks == 0) return 0;
    hipLaunchKernelGGL(k_adam_step, dim3(s.cl_blocks + fl_blocks), dim3(256), 0, (hipStream_t)stream, s);
    ADFP_CHECK_LAUNCH();
    return 0;
}
int adfp_prefilter_mask(const float* rays_o, const float* rays_d, const float* gt_depth, int n_rays, const double* bound_dev,
                        unsigned char* keep, float* depth_max, void* stream) {
    if (!rays_o || !rays_d || !gt_depth || !bound_dev || !keep || !depth_max || n_rays <= 0) return ADFP_E_ARG;
    hipLaunchKernelGGL(k_prefilter_mask, dim3(1), dim3(1024), 0, (hipStream_t)stream, rays_o, rays_d, gt_depth, n_rays, bound_dev, keep, depth_max);
    ADFP_CHECK_LAUNCH();
    return 0;
}
static int mapper_loss_impl(const adfp_loss_args* l, const AdamPrepArgs* prep, double* scratch, void* stream);
int adfp_mapper_loss(const adfp_loss_args* l, void* stream) { return mapper_loss_impl(l, nullptr, nullptr, stream); }
int adfp_mapper_loss_step(const adfp_loss_args* l, void* scratch, size_t scratch_bytes, int* steps, float* derived, int n_groups, const float* lr,
                          float beta1, float beta2, const int* skip_flag, void* stream) {
    if (!l || !scratch || ((uintptr_t)scratch & 7) || n_groups < 0 || n_groups > ADFP_ADAM_MAX_GROUPS) return ADFP_E_ARG;
    if (n_groups && (!steps || !derived || !lr)) return ADFP_E_ARG;
    if (l->n_rays > 0 && scratch_bytes < adfp_mapper_loss_scratch_bytes(l->n_rays)) return ADFP_E_WORKSPACE;
    AdamPrepArgs p; p.steps = steps; p.derived = derived; p.n = n_groups; p.beta1 = beta1; p.beta2 = beta2; p.skip = skip_flag;
    for (int g = 0; g < ADFP_ADAM_MAX_GROUPS; ++g) p.lr[g] = g < n_groups ? lr[g] : -1.f;
    return mapper_loss_impl(l, &p, (double*)scratch, stream);
}
size_t adfp_mapper_loss_scratch_bytes(int n_rays) { return n_rays < 0 ? 0 : (size_t)(1 + (n_rays + 255) / 256) * 8; }
static int mapper_loss_impl(const adfp_loss_args* l, const AdamPrepArgs* prep, double* scratch, void* stream) {
    if (!l || !l->depth || !l->gt_depth || !l->g_depth || !l->loss || l->n_rays < 0 || l->S <= 0) return ADFP_E_ARG;
    if (l->stage < ADFP_STAGE_LOW || l->stage > ADFP_STAGE_COLOR) return ADFP_E_ARG;
    if (l->stage == ADFP_STAGE_COLOR && (!l->color || !l->gt_color || !l->g_color)) return ADFP_E_ARG;
    if (l->warmup && (!l->weight || !l->g_weight)) return ADFP_E_ARG;
    if (l->n_rays == 0) return 0;
    LossArgs a;
    a.n = l->n_rays; a.S = l->S; a.color_term = l->stage == ADFP_STAGE_COLOR; a.warmup = l->warmup; a.w_color = l->w_color_loss;
    a.depth = l->depth; a.color = l->color; a.weight = l->weight; a.gt_depth = l->gt_depth; a.gt_color = l->gt_color; a.keep = l->keep;
    a.loss = l->loss; a.g_depth = l->g_depth; a.g_color = l->g_color; a.g_weight = l->g_weight;
    a.scratch = scratch; a.prep.n = 0;
    if (prep) a.prep = *prep;
    hipLaunchKernelGGL(k_mapper_loss, dim3((l->n_rays + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);      // one thread per ray
    ADFP_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
// ---- the Tracker iteration's head and tail, one workgroup each (adfp_tracker_head / adfp_tracker_tail) ----------------------------
__global__ __launch_bounds__(1024) void k_tracker_head(adfp_tracker_head_args a) {
    __shared__ float s_m[16];
    float c2w[16];
    camera_from_tensor_dev(a.cam, c2w);                     // every thread: 30 operations, the same values
    if (threadIdx.x < 16) a.c2w[threadIdx.x] = c2w[threadIdx.x];
    double b[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) b[k] = a.bound[k];
    const int Ww = a.W1 - a.W0;
    float mx = -INFINITY;
    for (int t = threadIdx.x; t < a.n; t += 1024) {
        // k_select_pixels
        const long long k = a.idx[t];
        const int row = a.H0 + (int)(k / Ww), col = a.W0 + (int)(k % Ww);
        const float pi = (float)col, pj = (float)row;
        a.pix_i[t] = pi; a.pix_j[t] = pj;
        const long long p = (long long)row * a.W + col;
        const float dep = a.depth_img[p];
        a.gt_depth[t] = dep;
        a.gt_color[3 * t] = a.color_img[3 * p]; a.gt_color[3 * t + 1] = a.color_img[3 * p + 1]; a.gt_color[3 * t + 2] = a.color_img[3 * p + 2];
        // k_rays_from_uv
        const float dx = (pi - a.cx) / a.fx, dy = -(pj - a.cy) / a.fy, dz = -1.f;
        float ro[3], rd[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            rd[m] = __fadd_rn(__fadd_rn(__fmul_rn(dx, c2w[4 * m + 0]), __fmul_rn(dy, c2w[4 * m + 1])), __fmul_rn(dz, c2w[4 * m + 2]));
            ro[m] = c2w[4 * m + 3];
            a.rays_d[3 * t + m] = rd[m]; a.rays_o[3 * t + m] = ro[m];
        }
        // k_prefilter_mask
        double tt = INFINITY; bool nan = false;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const double o = (double)ro[m], d = (double)rd[m];
            const double t0 = (b[2 * m] - o) / d, t1 = (b[2 * m + 1] - o) / d;
            nan |= (t0 != t0) | (t1 != t1);
            const double tm = t0 > t1 ? t0 : t1;
            tt = tm < tt ? tm : tt;
        }
        const bool k_ = !nan && (tt >= (double)dep);
        a.keep[t] = k_ ? 1 : 0;
        if (k_) mx = dep > mx ? dep : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float v = __shfl_xor(mx, o); mx = v > mx ? v : mx; }
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = s_m[0];
        for (int w = 1; w < 16; ++w) m = s_m[w] > m ? s_m[w] : m;
        *a.depth_max = m;
    }
}
__global__ __launch_bounds__(256) void k_tracker_tail(adfp_tracker_tail_args a) {
    __shared__ float s[4][12];
    __shared__ float s_g[16];
    // k_rays_from_uv_bwd
    float acc[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) acc[t] = 0.f;
    for (int i = threadIdx.x; i < a.n; i += 256) {
        const float dir[3] = {(a.pix_i[i] - a.cx) / a.fx, -(a.pix_j[i] - a.cy) / a.fy, -1.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float gd = a.g_rays_d ? a.g_rays_d[3 * i + k] : 0.f;
#pragma unroll
            for (int m = 0; m < 3; ++m) acc[4 * k + m] = fmaf(gd, dir[m], acc[4 * k + m]);
            acc[4 * k + 3] += a.g_rays_o ? a.g_rays_o[3 * i + k] : 0.f;
        }
    }
#pragma unroll
    for (int t = 0; t < 12; ++t) acc[t] = wave_sum(acc[t]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int t = 0; t < 12; ++t) s[threadIdx.x >> 6][t] = acc[t];
    __syncthreads();
    if (threadIdx.x < 16) {
        const float v = threadIdx.x < 12 ? (s[0][threadIdx.x] + s[1][threadIdx.x]) + (s[2][threadIdx.x] + s[3][threadIdx.x]) : 0.f;
        s_g[threadIdx.x] = v; a.g_c2w[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    // one thread from here on: 7 parameters
    float g_cam[7];
    camera_from_tensor_bwd_dev(a.cam, s_g, g_cam);
#pragma unroll
    for (int k = 0; k < 7; ++k) a.g_cam[k] = g_cam[k];
    if (!a.step) return;
    const bool better_before = *a.loss < *a.best_loss;
    if (a.n_groups == 2 && better_before) {                 // k_keep_best before the step
#pragma unroll
        for (int k = 0; k < 7; ++k) a.best_cam[k] = a.cam[k];
        *a.best_loss = *a.loss;
    }
    AdamPrepArgs pa;
    pa.steps = a.steps; pa.derived = a.derived; pa.n = a.n_groups; pa.beta1 = a.beta1; pa.beta2 = a.beta2; pa.skip = a.skip_flag;
    for (int g = 0; g < ADFP_ADAM_MAX_GROUPS; ++g) pa.lr[g] = g < a.n_groups ? a.lr[g] : -1.f;
    for (int g = 0; g < a.n_groups; ++g) adam_prep_group(pa, g);
    for (int g = 0; g < a.n_groups; ++g) {
        const int off = a.n_groups == 2 ? (g == 0 ? 4 : 0) : 0, cnt = a.n_groups == 2 ? (g == 0 ? 3 : 4) : 7;
        const float step_size = a.derived[2 * g], sqrt_bc2 = a.derived[2 * g + 1];
        if (sqrt_bc2 == 0.f) continue;                      // adam_prep_group's "skip this iteration"
        AdamArgs ad;
        ad.param = a.cam + off; ad.grad = a.g_cam + off; ad.exp_avg = a.exp_avg + off; ad.exp_avg_sq = a.exp_avg_sq + off; ad.mask = nullptr;
        ad.nvox = cnt; ad.C = 1; ad.beta1 = a.beta1; ad.beta2 = a.beta2; ad.eps = a.eps; ad.step_size = 0.f; ad.sqrt_bc2 = 1.f; ad.derived = nullptr;
        for (int k = 0; k < cnt; ++k) adam_one(ad, k, step_size, sqrt_bc2);
    }
    if (a.n_groups == 1 && better_before) {                 // k_keep_best after the step: the stepped pose, this iteration's loss
#pragma unroll
        for (int k = 0; k < 7; ++k) a.best_cam[k] = a.cam[k];
        *a.best_loss = *a.loss;
    }
}
extern "C" {
int adfp_tracker_head(const adfp_tracker_head_args* a, void* stream) {
    if (!a || a->n < 0 || a->H0 < 0 || a->W0 < 0 || a->H1 > a->H || a->W1 > a->W || a->H1 <= a->H0 || a->W1 <= a->W0) return ADFP_E_ARG;
    if (!a->cam || !a->c2w || !a->bound || !a->keep || !a->depth_max) return ADFP_E_ARG;
    if (a->n > 0 && (!a->idx || !a->depth_img || !a->color_img || !a->pix_i || !a->pix_j || !a->gt_depth || !a->gt_color || !a->rays_o || !a->rays_d)) return ADFP_E_ARG;
    hipLaunchKernelGGL(k_tracker_head, dim3(1), dim3(1024), 0, (hipStream_t)stream, *a);
    ADFP_CHECK_LAUNCH();
    return 0;
}
int adfp_tracker_tail(const adfp_tracker_tail_args* a, void* stream) {
    if (!a || a->n < 0 || !a->cam || !a->g_c2w || !a->g_cam) return ADFP_E_ARG;
    if (a->n > 0 && (!a->pix_i || !a->pix_j)) return ADFP_E_ARG;
    if (a->step && (!a->exp_avg || !a->exp_avg_sq || !a->steps || !a->derived || a->n_groups < 1 || a->n_groups > 2 || !a->loss || !a->best_loss || !a->best_cam))
        return ADFP_E_ARG;
    hipLaunchKernelGGL(k_tracker_tail, dim3(1), dim3(256), 0, (hipStream_t)stream, *a);
    ADFP_CHECK_LAUNCH();
    return 0;
}
int adfp_camera_from_tensor(const float* cam, float* c2w, void* stream) {
    if (!cam || !c2w) return ADFP_E_ARG;
    hipLaunchKernelGGL(k_camera_from_tensor, dim3(1), dim3(64), 0, (hipStream_t)stream, cam, c2w);
    ADFP_CHECK_LAUNCH();
    return 0;
}
int adfp_camera_from_tensor_backward(const float* cam, const float* g_c2w, float* g_cam, void* stream) {
    if (!cam || !g_c2w || !g_cam) return ADFP_E_ARG;
    hipLaunchKernelGGL(k_camera_from_tensor_bwd, dim3(1), dim3(64), 0, (hipStream_t)stream, cam, g_c2w, g_cam);
    ADFP_CHECK_LAUNCH();
    return 0;
}
int adfp_select_pixels(const long long* idx, int n, int H0, int H1, int W0, int W1, int H, int W, const float* depth_img, const float* color_img,
                       float* pix_i, float* pix_j, float* gt_depth, float* gt_color, void* stream) {
    if (n < 0 || H0 < 0 || W0 < 0 || H1 > H || W1 > W || H1 <= H0 || W1 <= W0) return ADFP_E_ARG;
    if (n == 0) return 0;
    if (!idx || !depth_img || !color_img || !pix_i || !pix_j || !gt_depth || !gt_color) return ADFP_E_ARG;
    hipLaunchKernelGGL(k_select_pixels, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, idx, n, H0, W0, W1 - W0, W, depth_img, color_img,
                       pix_i, pix_j, gt_depth, gt_color);
    ADFP_CHECK_LAUNCH();
    return 0;
}
int adfp_sample_keyframes(int n_frames, const adfp_keyframe* frames, int n, int H0, int H1, int W0, int W1, int H, int W, float fx, float fy, float cx,
                          float cy, float* rays_o, float* rays_d, float* gt_depth, float* gt_color, void* stream) {
    if (n_frames < 0 || n_frames > ADFP_KEYFRAMES_MAX || n < 0 || H0 < 0 || W0 < 0 || H1 > H || W1 > W || H1 <= H0 || W1 <= W0) return ADFP_E_ARG;
    if (n_frames == 0 || n == 0) return 0;
    if (!frames || !rays_o || !rays_d || !gt_depth || !gt_color) return ADFP_E_ARG;
    if ((long long)n_frames * n > 0x7fffffffll) return ADFP_E_UNSUPPORTED;
    KeyframeJobs a;
    for (int f = 0; f < ADFP_KEYFRAMES_MAX; ++f) {
        const adfp_keyframe& k = frames[f < n_frames ? f : 0];
        if (f < n_frames && (!k.idx || !k.depth_img || !k.color_img)) return ADFP_E_ARG;
        a.idx[f] = k.idx; a.c2w[f] = k.c2w; a.depth[f] = k.depth_img; a.color[f] = k.color_img;
        for (int m = 0; m < 12; ++m) a.pose[f][m] = k.c2w_host[m];
    }
    a.n_frames = n_frames; a.n = n; a.H0 = H0; a.W0 = W0; a.Ww = W1 - W0; a.W = W; a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy;
    a.ro = rays_o; a.rd = rays_d; a.gd = gt_depth; a.gc = gt_color;
    hipLaunchKernelGGL(k_sample_keyframes, dim3((n_frames * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    ADFP_CHECK_LAUNCH();
    return 0;
}
int adfp_track_keep_best(const double* loss, const float* cam, double* best_loss, float* best_cam, void* stream) {
    if (!loss || !cam || !best_loss || !best_cam) return ADFP_E_ARG;
    hipLaunchKernelGGL(k_keep_best, dim3(1), dim3(64), 0, (hipStream_t)stream, loss, cam, best_loss, best_cam);
    ADFP_CHECK_LAUNCH();
    return 0;
}
int adfp_tracker_loss(const adfp_track_loss_args* l, void* stream) {
    if (!l || !l->depth || !l->uncertainty || !l->color || !l->gt_depth || !l->gt_color || !l->g_depth || !l->g_color || !l->loss || l->n_rays < 0) return ADFP_E_ARG;
    if (l->n_rays > ADFP_TRACK_MAX_RAYS) return ADFP_E_UNSUPPORTED;
    TrackLossArgs a;
    a.n = l->n_rays; a.handle_dynamic = l->handle_dynamic; a.w_color = l->w_color_loss;
    a.depth = l->depth; a.unc = l->uncertainty; a.color = l->color; a.gd = l->gt_depth; a.gc = l->gt_color; a.keep = l->keep;
    a.loss = l->loss; a.g_depth = l->g_depth; a.g_color = l->g_color;
    hipLaunchKernelGGL(k_tracker_loss, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    ADFP_CHECK_LAUNCH();
    return 0;
}

int adfp_ray_sort_keys(const float* rays_o, const float* rays_d, const float* gt_depth, int n_rays, const double tsdf_bnds[3][2], int* key, int* val,
                       void* stream) {
    if (!rays_o || !rays_d || !tsdf_bnds || !key || !val || n_rays < 0) return ADFP_E_ARG;
    if (n_rays == 0) return 0;
    RayKeyArgs a; a.ro = rays_o; a.rd = rays_d; a.gd = gt_depth; a.n = n_rays; a.key = key; a.val = val;
    for (int k = 0; k < 3; ++k) {
        if (!(tsdf_bnds[k][1] > tsdf_bnds[k][0])) return ADFP_E_ARG;
        a.lo[k] = (float)tsdf_bnds[k][0]; a.inv[k] = (float)(1.0 / (tsdf_bnds[k][1] - tsdf_bnds[k][0]));
    }
    hipLaunchKernelGGL(k_ray_sort_keys, dim3((n_rays + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    ADFP_CHECK_LAUNCH();
    return 0;
}
int adfp_ray_order_probe(const float* rays_o, const float* rays_d, const float* gt_depth, int n_rays, float far_distance, int* verdict, void* stream) {
    if (!rays_o || !rays_d || !verdict || n_rays < 0 || !(far_distance > 0.f)) return ADFP_E_ARG;
    hipLaunchKernelGGL(k_ray_order_probe, dim3(1), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, gt_depth, n_rays, far_distance * far_distance, verdict);
    ADFP_CHECK_LAUNCH();
    return 0;
}
int adfp_gather_pack(int n_arrays, const void* const* src, const int* words, long long rows, void* dst, void* stream) {
    if (n_arrays < 1 || n_arrays > ADFP_GATHER_MAX || !src || !words || !dst || rows < 0) return ADFP_E_ARG;
    GatherJobs j; j.n = n_arrays; j.world = 1; j.pad = rows; j.prefix[0] = 0; j.prefix[1] = rows; j.woff[0] = 0;
    for (int a = 0; a < n_arrays; ++a) {
        if (!src[a] || words[a] <= 0) return ADFP_E_ARG;
        j.arr[a] = (unsigned*)src[a]; j.words[a] = words[a]; j.woff[a + 1] = j.woff[a] + words[a];
    }
    const long long total = rows * j.woff[n_arrays];
    if (total == 0) return 0;
    if (total > 0x7fffffffll * 256) return ADFP_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_gather_rows<true>, dim3((unsigned)((total + 255) / 256), 1), dim3(256), 0, (hipStream_t)stream, j, (unsigned*)dst);
    ADFP_CHECK_LAUNCH();
    return 0;
}
int adfp_gather_unpack(int n_arrays, void* const* dst, const int* words, int world, long long pad, const long long* rows_per_rank,
                       const void* gathered, void* stream) {
    if (n_arrays < 1 || n_arrays > ADFP_GATHER_MAX || !dst || !words || !rows_per_rank || !gathered || world < 1 || world > ADFP_GATHER_MAX_RANKS || pad < 0)
        return ADFP_E_ARG;
    GatherJobs j; j.n = n_arrays; j.world = world; j.pad = pad; j.prefix[0] = 0; j.woff[0] = 0;
    for (int r = 0; r < world; ++r) {
        if (rows_per_rank[r] < 0 || rows_per_rank[r] > pad) return ADFP_E_ARG;
        j.prefix[r + 1] = j.prefix[r] + rows_per_rank[r];
    }
    for (int a = 0; a < n_arrays; ++a) {
        if (!dst[a] || words[a] <= 0) return ADFP_E_ARG;
        j.arr[a] = (unsigned*)dst[a]; j.words[a] = words[a]; j.woff[a + 1] = j.woff[a] + words[a];
    }
    const long long per_rank = pad * j.woff[n_arrays];
    if (per_rank == 0) return 0;
    if (per_rank > 0x7fffffffll * 256) return ADFP_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_gather_rows<false>, dim3((unsigned)((per_rank + 255) / 256), (unsigned)world), dim3(256), 0, (hipStream_t)stream, j, (unsigned*)gathered);
    ADFP_CHECK_LAUNCH();
    return 0;
}

int adfp_composite(const float* raw, const double* z_vals, int n_rays, int S, double* depth, double* uncertainty, float* color,
                   float* weights, void* stream) {
    if (!raw || !z_vals || !depth || !uncertainty || !color || n_rays < 0 || S <= 0) return ADFP_E_ARG;
    if (n_rays == 0) return 0;
    hipLaunchKernelGGL(k_composite, dim3((n_rays + 3) / 4), dim3(256), 0, (hipStream_t)stream, raw, z_vals, n_rays, S, depth,
                       uncertainty, color, weights);
    ADFP_CHECK_LAUNCH();
    return 0;
}

int adfp_render_forward(const adfp_scene* scene, const adfp_render_args* r, void* stream) {
    if (!r) return ADFP_E_ARG;
    int rc = check_scene(scene, r->stage); if (rc) return rc;
    const adfp_frame_job* fr = r->frame;
    if (fr) {          // the call's rays are pixels [first, first + n_rays) of a frame: rays and far clamps come from the frame
        if (!fr->c2w || !fr->depth || !fr->rays_o || !fr->rays_d || fr->H <= 0 || fr->W <= 0 || (long long)fr->H * fr->W > 0x7fffffffll) return ADFP_E_ARG;
        if (r->depth_max || r->depth_max_segment <= 0 || r->depth_max_first_ray < 0) return ADFP_E_ARG;
        if ((long long)r->depth_max_first_ray + r->n_rays > (long long)fr->H * fr->W) return ADFP_E_ARG;
        if (r->perturb > 0.f) return ADFP_E_UNSUPPORTED;
    }
    const float* rays_o = fr ? fr->rays_o : r->rays_o;
    const float* rays_d = fr ? fr->rays_d : r->rays_d;
    const float* gt_depth = fr ? fr->depth + r->depth_max_first_ray : r->gt_depth;
    if (!rays_o || !rays_d || !r->depth || !r->uncertainty || !r->color || !r->weight || !r->workspace) return ADFP_E_ARG;
    if (r->n_rays < 0 || r->n_samples <= 0 || r->n_surface < 0 || r->depth_max_segment < 0) return ADFP_E_ARG;
    if (r->depth_max_first_ray < 0 || (r->depth_max_first_ray > 0 && !fr && (!r->depth_max || r->depth_max_segment <= 0 || !gt_depth))) return ADFP_E_ARG;
    if (r->n_pack_jobs < 0 || r->n_pack_jobs > ADFP_PACK_MAX_JOBS || (r->n_pack_jobs && !r->pack_jobs)) return ADFP_E_ARG;
    if (r->n_relayout_jobs < 0 || r->n_relayout_jobs > ADFP_RELAYOUT_MAX_JOBS || (r->n_relayout_jobs && !r->relayout_jobs)) return ADFP_E_ARG;
    if ((r->prefilter_bound != nullptr) != (r->prefilter_keep != nullptr)) return ADFP_E_ARG;
    if (r->prefilter_bound && (fr || !gt_depth || r->depth_max || r->depth_max_segment != 0 || r->depth_max_first_ray != 0)) return ADFP_E_ARG;
    if (r->state && r->stage != ADFP_STAGE_LOW &&
        (!r->state->flags || !r->state->list || !r->state->counter || !r->state->att_occ || !r->state->att_u)) return ADFP_E_ARG;
    const int S = r->n_samples + (gt_depth ? r->n_surface : 0);
    if (S > ADFP_MAX_SAMPLES) return ADFP_E_UNSUPPORTED;
    const long long Pn = (long long)r->n_rays * S;
    if (Pn > 0x7fffffffll) return ADFP_E_UNSUPPORTED;
    // per-segment far clamp: the partial maxima live in SEGMAX_SEGS x SEGMAX_PARTS words of the workspace head (refused here, before anything is launched)
    if (gt_depth && !r->depth_max && r->depth_max_segment > 0 &&
        ((fr ? (long long)fr->H * fr->W : (long long)r->n_rays) + r->depth_max_segment - 1) / r->depth_max_segment > SEGMAX_SEGS)
        return ADFP_E_UNSUPPORTED;
    Workspace ws = carve(r->workspace, Pn);
    if (r->workspace_bytes < ws.bytes) return ADFP_E_WORKSPACE;
    if (r->n_rays == 0) {                                // the images and the grid conversions are still owed
        rc = r->n_pack_jobs ? adfp_pack_images(r->n_pack_jobs, r->pack_jobs, scene->status, stream) : 0;
        return rc ? rc : adfp_relayout_grids(r->n_relayout_jobs, r->relayout_jobs, 0, stream);
    }
    hipStream_t st = (hipStream_t)stream;
    double* z = r->z_vals ? r->z_vals : ws.z;
    float* raw = r->raw ? r->raw : ws.raw;
    // The call's FIRST launch (k_forward_head) does everything that has to precede the sampler:
    //  * ONE zero fill for the call's small device words: the in-band counter and range flag (bytes 0-63) and the tile counters
    //    share the first 256 bytes of the workspace -- and, in a training call, the caller's counter block that eval_points_impl
    //    uses instead;
    //  * the weight images the caller handed over to pack;
    //  * a frame job's rays;
    //  * the partial maxima of gt_depth (per segment of rays, or one "segment" = the whole call) when no depth_max was given.
    ForwardHeadArgs h;
    memset(&h, 0, sizeof(h));
    {
        ZeroBatch zb;
        hipError_t e = zb.add(ws.counter, 256, st);
        if (e == hipSuccess && r->state && r->state->counter && r->state->counter != ws.counter) e = zb.add(r->state->counter, 64, st);
        if (e != hipSuccess) return (int)e;
        h.z = zb.z; h.nb_zero = (int)zb.blocks;
    }
    h.p.n = 0; h.nb_pack = 0;
    if (r->n_pack_jobs > 0) {
        rc = pack_jobs_table(r->n_pack_jobs, r->pack_jobs, scene->status, h.p); if (rc) return rc;
        h.nb_pack = h.p.first[h.p.n];
    }
    h.nb_rays = 0;
    if (fr) {
        h.rj.c2w = fr->c2w; h.rj.W = fr->W; h.rj.fx = fr->fx; h.rj.fy = fr->fy; h.rj.cx = fr->cx; h.rj.cy = fr->cy;
        h.rj.first = r->depth_max_first_ray; h.rj.n = r->n_rays; h.rj.ro = fr->rays_o; h.rj.rd = fr->rays_d;
        h.nb_rays = (r->n_rays + 255) / 256;
    }
    const unsigned* seg_parts = nullptr;
    int segment = r->depth_max_segment;
    unsigned nb_seg = 0;
    if (gt_depth && !r->depth_max) {
        // whole frame's depth (frame job) or the call's own rays; no segments = one segment that is the whole call
        h.sj.d = fr ? fr->depth : gt_depth;
        h.sj.n = fr ? fr->H * fr->W : r->n_rays;
        h.sj.seg = segment > 0 ? segment : h.sj.n;
        h.sj.nseg = (h.sj.n + h.sj.seg - 1) / h.sj.seg;
        h.sj.parts = ws.segparts;
        h.sj.keep = r->prefilter_keep; h.sj.bnd = r->prefilter_bound; h.sj.ro = rays_o; h.sj.rd = rays_d;       // NULL: a plain maximum
        if (segment <= 0) segment = h.sj.n;                // k_sample indexes segment (first + ray) / segment = 0
        nb_seg = (unsigned)h.sj.nseg * SEGMAX_PARTS;
        seg_parts = ws.segparts;
    }
    RelayoutJobs rl; rl.n = 0;                          // the grids the caller handed over: converted by blocks of the SAMPLER's launch
    if (r->n_relayout_jobs > 0) { rc = relayout_jobs_table(r->n_relayout_jobs, r->relayout_jobs, rl); if (rc) return rc; }
    hipLaunchKernelGGL(k_forward_head, dim3((unsigned)(h.nb_pack + h.nb_zero + h.nb_rays) + nb_seg), dim3(256), 0, st, h);
    ADFP_CHECK_LAUNCH();
    rc = sample_rays_impl(rays_o, rays_d, gt_depth, r->n_rays, scene->bound, r->n_samples, r->n_surface, r->lindisp,
                          r->perturb, r->t_rand, r->depth_max, z, (char*)ws.counter + 64, stream, true, segment, r->depth_max_first_ray, seg_parts, &rl);
    if (rc) return rc;
    PtsDev P;
    P.mode = ADFP_PTS_RAYS; P.S = S; P.n = (int)Pn; P.pts = nullptr; P.ro = rays_o; P.rd = rays_d; P.z = z;
    rc = eval_points_impl(scene, P, r->stage, 1, raw, r->weight, ws, st, r->state, true);
    if (rc) return rc;
    return adfp_composite(raw, z, r->n_rays, S, r->depth, r->uncertainty, r->color, nullptr, stream);
}


}  // extern "C"

// ---------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------
// Rows of the staging buffer (the widest row: the attention network's, 3.3 KB).  The list-indexed kernels (attention backward, its
// weight gradients) are launched once per chunk of 2 x this many LIST entries -- the host cannot know how many entries the in-band
// list has, so chunks beyond its end are launches that return at once (~5 us each inside a graph replay).  160 K rows make a
// 5 000-ray x 64-sample iteration (320 000 points) ONE chunk: four launches fewer per iteration than with 64 K rows; 545 MB.
#define STG_ROWS_MAX 163840

#define OUTER_NSLOT 256         // workgroups per weight-gradient launch = private gradient copies (one per CU); outer_slots() = the launch's cap
struct BwdWorkspace { float* g_raw; float* att_g; float* g_pts; float* stage; int stage_rows; float* partial; int part_stride; float* gmax; float* gmax_parts;
                      int gmax_pending;              // > 0: gmax_parts[0 .. gmax_pending) still wait to be folded into gmax (backward_points)
                      // adfp_render_backward's first two launches (zero fill of the gradient outputs, k_composite_bwd), held back so that
                      // backward_points can send them off in ONE launch with k_bin_keys (k_backward_head)
                      bool head_zeroes_g_pts;        // ... and g_pts with them
                      float* g_pts_dec; size_t g_pts_stride;   // [ADFP_PGRAD_MAX_JOBS][P,3]: the decoders' own d/d position buffers (k_decode_bwd_h_pgrad3)
                      bool pgrad_separate;           // adfp_render_backward: the decoders write those and k_rays_grad adds them up
                      int pgrad_used;                // how many of them the call's launch wrote
                      bool head_pending; ZeroJobs head_zero; unsigned head_zero_blocks; CompositeBwdArgs head_comp;
                      // adfp_backward_args.side_stream / side_events (host objects of the caller): the sort's own lane; join_pending =
                      // the main stream has not waited for the sort's end yet (flush_scatter does)
                      hipStream_t side; hipEvent_t side_ev[2]; bool side_join_pending;
                      float* gc; size_t gc_stride; int* bin_key; int* bin_val; int* bin_key_sorted; int* bin_perm; int* sort_table;
                      size_t bytes; };
static BwdWorkspace carve_bwd(void* base, long long P) {
    BwdWorkspace w; size_t o = 0;
    w.gmax_pending = 0; w.head_pending = false; w.head_zeroes_g_pts = false; w.pgrad_separate = false; w.pgrad_used = 0;
    w.side = nullptr; w.side_ev[0] = w.side_ev[1] = nullptr; w.side_join_pending = false;
    w.g_raw = at<float>(base, o); o += align256((size_t)P * 16);
    w.att_g = at<float>(base, o); o += align256((size_t)P * 4);
    w.g_pts = at<float>(base, o); o += align256((size_t)P * 12);
    w.g_pts_stride = align256((size_t)P * 12) / 4;
    w.g_pts_dec = at<float>(base, o); o += ADFP_PGRAD_MAX_JOBS * w.g_pts_stride * 4;
    w.stage_rows = (int)(P < STG_ROWS_MAX ? P : STG_ROWS_MAX);
    if (w.stage_rows < 32) w.stage_rows = 32;
    w.stage = at<float>(base, o); o += align256((size_t)w.stage_rows * AttStage::NCOLS * 4);
    int fmax = DecLayout<32, 1>::F_TOTAL;
    if (DecLayout<64, 1>::F_TOTAL > fmax) fmax = DecLayout<64, 1>::F_TOTAL;
    if (DecLayout<32, 4>::F_TOTAL > fmax) fmax = DecLayout<32, 4>::F_TOTAL;
    if (AttLayout::F_TOTAL > fmax) fmax = AttLayout::F_TOTAL;
    w.part_stride = (fmax + 63) / 64 * 64;
    w.partial = at<float>(base, o); o += align256((size_t)OUTER_NSLOT * w.part_stride * 4);
    w.gmax = at<float>(base, o); o += 256;            // largest |cotangent of raw| of the call (grad_scale)
    w.gmax_parts = at<float>(base, o); o += align256((size_t)P * 4);     // ... per ray / per workgroup, before k_max_reduce
    // spatially ordered grid-gradient scatter of the f16-split backward (k_scatter_sorted): d/d c rows, sort keys, sorted order
    w.gc_stride = align256((size_t)P * 128) / 4;                          // one [P][32] block of rows per grid (ADFP_SCATTER_MAX_JOBS of them)
    w.gc = at<float>(base, o); o += ADFP_SCATTER_MAX_JOBS * w.gc_stride * 4;
    w.bin_key = at<int>(base, o); o += align256((size_t)P * 4);
    w.bin_val = at<int>(base, o); o += align256((size_t)P * 4);
    w.bin_key_sorted = at<int>(base, o); o += align256((size_t)P * 4);
    w.bin_perm = at<int>(base, o); o += align256((size_t)P * 4);
    w.sort_table = at<int>(base, o); o += align256(((((size_t)P + ADFP_RS_TILE - 1) / ADFP_RS_TILE) + 1) * ADFP_RS_DIGITS * 4);     // [digits][tiles] + [digits] totals
    w.bytes = o;
    return w;
}
extern "C" size_t adfp_backward_workspace_bytes(long long n_points) {
    if (n_points < 0) return 0;
    return carve_bwd(nullptr, n_points).bytes;
}

static void add_job(OuterArgs& a, int colA, int colB, int dst, int rs, int cs, int nr, int j0, int nc) {
    OuterJob j; j.colA = colA; j.colB = colB; j.dst = dst; j.rs = rs; j.cs = cs; j.nr = nr; j.j0 = j0; j.nc = nc;
    a.jobs[a.njobs++] = j;
}
template <int CDIM, int NOUT>
static void decoder_jobs(OuterArgs& a) {
    using L = DecLayout<CDIM, NOUT>;
    using ST = DecStage<CDIM>;
    a.njobs = 0; a.ncols = ST::NCOLS;
    for (int i = 0; i < 5; ++i) {
        const int ind = L::in_dim(i);
        if (i == 0 || i == 3) {
            for (int b = 0; b < 3; ++b) add_job(a, ST::SGP(i), ST::SE + 32 * b, L::F_PL(i) + 32 * b, ind, 1, 32, 0, b < 2 ? 32 : 29);
            if (i == 3) add_job(a, ST::SGP(3), ST::SH(2), L::F_PL(3) + 93, ind, 1, 32, 0, 32);
        } else add_job(a, ST::SGP(i), ST::SH(i - 1), L::F_PL(i), ind, 1, 32, 0, 32);
        add_job(a, ST::SGP(i), ST::SX, L::F_PL(i) + 32 * ind, 1, 0, 32, 3, 1);                 // bias: column "1" of X
        for (int cb = 0; cb < CDIM / 32; ++cb) add_job(a, ST::SGH(i), ST::SC + 32 * cb, L::F_FC(i) + 32 * cb, CDIM, 1, 32, 0, 32);
        add_job(a, ST::SGH(i), ST::SX, L::F_FC(i) + 32 * CDIM, 1, 0, 32, 3, 1);
    }
    for (int b = 0; b < 3; ++b) add_job(a, ST::SGA + 32 * b, ST::SX, L::F_EB + 32 * b, 1, 93, b < 2 ? 32 : 29, 0, 3);   // embedder._B [3][93]
    add_job(a, ST::SGO, ST::SH(4), L::F_OW, 32, 1, NOUT, 0, 32);
    add_job(a, ST::SGO, ST::SX, L::F_OB, 1, 0, NOUT, 3, 1);
}
static void attention_jobs(OuterArgs& a) {
    using A = AttLayout;
    using ST = AttStage;
    a.njobs = 0; a.ncols = ST::NCOLS;
    for (int ob = 0; ob < 2; ++ob) {
        add_job(a, ST::AG0 + 32 * ob, ST::AX, A::F_W0 + 64 * ob, 2, 1, 32, 0, 2);
        add_job(a, ST::AG0 + 32 * ob, ST::AX, A::F_B0 + 32 * ob, 1, 0, 32, 2, 1);
    }
    for (int ob = 0; ob < 4; ++ob) {
        for (int ib = 0; ib < 2; ++ib) add_job(a, ST::AG1 + 32 * ob, ST::AH0 + 32 * ib, A::F_W1 + 32 * ob * 64 + 32 * ib, 64, 1, 32, 0, 32);
        add_job(a, ST::AG1 + 32 * ob, ST::AX, A::F_B1 + 32 * ob, 1, 0, 32, 2, 1);
        for (int ib = 0; ib < 4; ++ib) add_job(a, ST::AG2 + 32 * ob, ST::AH1 + 32 * ib, A::F_W2 + 32 * ob * 128 + 32 * ib, 128, 1, 32, 0, 32);
        add_job(a, ST::AG2 + 32 * ob, ST::AX, A::F_B2 + 32 * ob, 1, 0, 32, 2, 1);
    }
    for (int ob = 0; ob < 2; ++ob) {
        for (int ib = 0; ib < 4; ++ib) add_job(a, ST::AG3 + 32 * ob, ST::AH2 + 32 * ib, A::F_W3 + 32 * ob * 128 + 32 * ib, 128, 1, 32, 0, 32);
        add_job(a, ST::AG3 + 32 * ob, ST::AX, A::F_B3 + 32 * ob, 1, 0, 32, 2, 1);
        add_job(a, ST::AGL, ST::AH3 + 32 * ob, A::F_WO + 32 * ob, 64, 1, 2, 0, 32);
    }
    add_job(a, ST::AGL, ST::AX, A::F_BO, 1, 0, 2, 2, 1);
}

// weight gradients of one network: outer_begin (zero the per-workgroup copies), launch_outer per staging chunk,
// outer_end (sum the copies into the flat gradient)
static int outer_begin(const BwdWorkspace& bw, int n_floats, hipStream_t st) {
#ifdef ADFP_OUTER_SIMPLE
    return 0;
#else
    (void)n_floats;
    return (int)zero_async(bw.partial, (size_t)OUTER_NSLOT * bw.part_stride * 4, st);
#endif
}
static int outer_end(const BwdWorkspace& bw, int n_floats, float* flat, hipStream_t st) {
#ifndef ADFP_OUTER_SIMPLE
    hipLaunchKernelGGL(k_reduce_partials, dim3((n_floats + 255) / 256), dim3(256), 0, st, bw.partial, OUTER_NSLOT, bw.part_stride, n_floats, flat);
    ADFP_CHECK_LAUNCH();
#endif
    return 0;
}
static int outer_end_scaled(const BwdWorkspace& bw, int n_floats, float* flat, hipStream_t st) {
    hipLaunchKernelGGL(k_reduce_partials_scaled, dim3((n_floats + 31) / 32), dim3(256), 0, st, bw.partial, OUTER_NSLOT, bw.part_stride, n_floats, flat, bw.gmax);
    ADFP_CHECK_LAUNCH();
    return 0;
}
static int launch_outer(OuterArgs& oa, const BwdWorkspace& bw, const int* count_ptr, int lo, int hi, float* flat, hipStream_t st) {
    const float* stage = bw.stage;
    oa.stage = stage; oa.count_ptr = count_ptr; oa.chunk_lo = lo; oa.chunk_hi = hi; oa.flat = flat;
    oa.partial = bw.partial; oa.part_stride = bw.part_stride;
    const int rows = hi - lo;
#ifdef ADFP_OUTER_SIMPLE       // A/B switch: one wave per (job, 512 rows), operands straight from L2
    oa.rows_per_wave = 512;
    hipLaunchKernelGGL(k_outer, dim3((rows + oa.rows_per_wave - 1) / oa.rows_per_wave, oa.njobs), dim3(64), 0, st, oa);
#else
    if (oa.ncols > OUTER_MAXCOLS || (oa.ncols & 3) || oa.njobs > OUTER_NW * OUTER_JW) return ADFP_E_UNSUPPORTED;
    // rows per workgroup: at most OUTER_NSLOT workgroups (each owns one private gradient copy), at least 64 rows each
    int per = (rows + OUTER_NSLOT - 1) / OUTER_NSLOT;
    per = ((per < 64 ? 64 : per) + OUTER_RT - 1) / OUTER_RT * OUTER_RT;
    oa.rows_per_wave = per;
    hipLaunchKernelGGL(k_outer_lds, dim3((rows + per - 1) / per), dim3(512), 0, st, oa);
#endif
    ADFP_CHECK_LAUNCH();
    return 0;
}

template <int CDIM, int NOUT, int ROLE, bool PGRAD>
static int run_decode_bwd_p(DecodeBwdArgs a, int total, const int* count_ptr, float* flat, BwdWorkspace& bw, hipStream_t st) {
    if (total == 0) return 0;
    // the scatter cache keeps its slot number in the top 5 bits of the voxel index
    if (ROLE != ROLE_HIGH && a.g_grid && (long long)a.g0.X * a.g0.Y * a.g0.Z >= (1ll << 27)) return ADFP_E_UNSUPPORTED;
    if (!flat) {
        a.stage = nullptr; a.chunk_lo = 0; a.chunk_hi = total;
        const int ntiles = (total + 31) / 32;
        hipLaunchKernelGGL((k_decode_bwd<CDIM, NOUT, ROLE, false, PGRAD, 256>), dim3(decode_grid(ntiles, 4, 1)), dim3(256), 0, st, a);
        ADFP_CHECK_LAUNCH();
        return 0;
    }
    OuterArgs oa; decoder_jobs<CDIM, NOUT>(oa);
    a.stage = bw.stage;
    const int rows_cap = (int)((size_t)bw.stage_rows * AttStage::NCOLS / DecStage<CDIM>::NCOLS);
    int rc = outer_begin(bw, DecLayout<CDIM, NOUT>::F_TOTAL, st);
    if (rc) return rc;
    for (int lo = 0; lo < total; lo += rows_cap) {
        const int hi = lo + rows_cap < total ? lo + rows_cap : total;
        a.chunk_lo = lo; a.chunk_hi = hi;
        const int ntiles = (hi - lo + 31) / 32;
        hipLaunchKernelGGL((k_decode_bwd<CDIM, NOUT, ROLE, true, PGRAD, 256>), dim3(decode_grid(ntiles, 4, 1)), dim3(256), 0, st, a);
        ADFP_CHECK_LAUNCH();
        rc = launch_outer(oa, bw, count_ptr, lo, hi, flat, st);
        if (rc) return rc;
    }
    return outer_end(bw, DecLayout<CDIM, NOUT>::F_TOTAL, flat, st);
}
// The f16-split backward of one decoder (adfp_backward_h.h): `t` = its T image, `masks` / `act` = what the training forward
// left.  Weight gradients: the G part of the staging rows is chunked like the exact path's rows (so that a chunk is still in
// the Infinity Cache when k_outer_h reads it back); the X part lies in `act` for all rows.
#define ADFP_BWDH_NT 384
#ifdef ADFP_TUNE_ROLE_SHARES       // tuning builds only (the product library reads no environment): ADFP_ROLE_SHARES="<P>,<H>" of 256
static int role_share_env(int which, int dflt) {
    const char* e = getenv("ADFP_ROLE_SHARES");
    int v[2];
    if (!e || sscanf(e, "%d,%d", &v[0], &v[1]) != 2 || v[0] < 1 || v[1] < 1 || v[0] + v[1] > 254) return dflt;
    return v[which];
}
#else
static constexpr int role_share_env(int, int dflt) { return dflt; }
#endif
// sort of the call's points for k_scatter_sorted (set up once per backward call by backward_points)
// Sorts n (key, value) pairs by the low key_bits bits of the key, stable.  The two buffer pairs are used in turn; *key_fin / *val_fin
// = the pair the last pass wrote (a / b).  table: ADFP_RS_DIGITS * (ceil(n / ADFP_RS_TILE) + 1) ints.
static int radix_sort_pairs(int* key_a, int* val_a, int* key_b, int* val_b, int n, int key_bits, int* table, const int** key_fin,
                            const int** val_fin, hipStream_t st, const float* max_parts = nullptr, int max_n = 0, float* max_out = nullptr,
                            hipEvent_t after_first_launch = nullptr) {
    RadixArgs rs; rs.table = table; rs.n = n; rs.ntiles = (n + ADFP_RS_TILE - 1) / ADFP_RS_TILE;
    rs.max_parts = nullptr; rs.max_n = 0; rs.max_out = nullptr;
    constexpr int db = ADFP_RS_DIGIT_BITS, nd = ADFP_RS_DIGITS;
    rs.totals = table + (size_t)nd * rs.ntiles;
    const int passes = (key_bits + db - 1) / db;
    int* kin = key_a; int* vin = val_a; int* kout = key_b; int* vout = val_b;
    for (int ps = 0; ps < passes; ++ps) {
        rs.key_in = kin; rs.val_in = vin; rs.key_out = kout; rs.val_out = vout; rs.shift = db * ps;
        if (ps == 0 && max_parts) { rs.max_parts = max_parts; rs.max_n = max_n; rs.max_out = max_out; }      // one more workgroup: the fold
        hipLaunchKernelGGL(k_rs_hist<db>, dim3(rs.ntiles + (rs.max_parts ? 1 : 0)), dim3(256), 0, st, rs);
        ADFP_CHECK_LAUNCH();
        if (ps == 0 && after_first_launch) { hipError_t e = hipEventRecord(after_first_launch, st); if (e != hipSuccess) return (int)e; }
        rs.max_parts = nullptr;
        hipLaunchKernelGGL(k_rs_scan, dim3(nd / 4), dim3(256), 0, st, rs.table, rs.ntiles, rs.totals);
        ADFP_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_rs_scatter<db>, dim3(rs.ntiles), dim3(256), 0, st, rs);
        ADFP_CHECK_LAUNCH();
        int* tk = kin; kin = kout; kout = tk; int* tv = vin; vin = vout; vout = tv;
    }
    if (key_fin) *key_fin = kin;
    if (val_fin) *val_fin = vin;
    return 0;
}
extern "C" size_t adfp_sort_workspace_bytes(long long n) { return n < 0 ? 0 : ((size_t)((n + ADFP_RS_TILE - 1) / ADFP_RS_TILE) + 1) * ADFP_RS_DIGITS * 4; }
extern "C" int adfp_sort_pairs(int* key, int* val, int* key_tmp, int* val_tmp, long long n, int key_bits, void* workspace, size_t workspace_bytes,
                               void* stream) {
    if (!key || !val || !key_tmp || !val_tmp || !workspace || n < 0 || key_bits < 1 || key_bits > 31) return ADFP_E_ARG;
    if (n > 0x7fffffffll - ADFP_RS_TILE) return ADFP_E_UNSUPPORTED;      // the tile arithmetic of the sort kernels is int
    if (workspace_bytes < adfp_sort_workspace_bytes(n)) return ADFP_E_WORKSPACE;
    if (n == 0) return 0;
    const int* kf; const int* vf;
    int rc = radix_sort_pairs(key, val, key_tmp, val_tmp, (int)n, key_bits, (int*)workspace, &kf, &vf, (hipStream_t)stream);
    if (rc) return rc;
    if (kf != key) {                               // an odd number of passes: bring the result home
        hipError_t e = hipMemcpyAsync(key, kf, (size_t)n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        if (e == hipSuccess) e = hipMemcpyAsync(val, vf, (size_t)n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

// The head of a backward call in ONE launch: the zero fill of the gradient outputs, the compositing backward and the sort keys are
// independent of each other (three launches of 5-7 us each inside a graph replay).  Workgroups [0, nb_comp) = k_composite_bwd's,
// then nb_bin of k_bin_keys', then the zero fill's; nb_bin may be 0 (no sorted scatter in the call).
struct BwdHeadArgs { CompositeBwdArgs c; BinArgs b; ZeroJobs z; int nb_comp, nb_bin; };
__global__ __launch_bounds__(256) void k_backward_head(BwdHeadArgs h) {
    int blk = (int)blockIdx.x;
    if (blk < h.nb_comp) { composite_bwd_block(h.c, blk); return; }
    blk -= h.nb_comp;
    if (blk < h.nb_bin) { bin_keys_block(h.b, blk); return; }
    zero_multi_block(h.z, (unsigned)(blk - h.nb_bin));
}
static int launch_backward_head(BwdWorkspace& bw, const BinArgs* bins, int P, hipStream_t st) {
    BwdHeadArgs h;
    h.c = bw.head_comp; h.z = bw.head_zero;
    h.nb_comp = (bw.head_comp.n_rays + 3) / 4;
    h.nb_bin = bins ? (P + 255) / 256 : 0;
    if (bins) h.b = *bins; else memset(&h.b, 0, sizeof(h.b));
    h.b.max_parts = nullptr;
    hipLaunchKernelGGL(k_backward_head, dim3(h.nb_comp + h.nb_bin + bw.head_zero_blocks), dim3(256), 0, st, h);
    ADFP_CHECK_LAUNCH();
    bw.head_pending = false;
    return 0;
}

// the grids whose d/d c rows wait for the call's ONE k_scatter_sorted launch (flush_scatter): job k has rows in bw.gc + k * gc_stride
struct BinPlan { bool ok; BinArgs args; const int* perm; ScatterMultiArgs pend;
                 DecodeBwdH3Args pgrad; };            // the position-gradient backwards of the call's decoders, launched together (flush_pgrad)
static int scatter_bins(BinPlan& bp, const DecodeBwdArgs& o, float* gc, const unsigned char* flags) {
    if (o.g0.X > 1023 || o.g0.Y > 1023 || o.g0.Z > 1023) return ADFP_E_UNSUPPORTED;          // packed cell coordinates
    if (bp.pend.n_jobs >= ADFP_SCATTER_MAX_JOBS) return ADFP_E_UNSUPPORTED;
    ScatterSortedArgs& s = bp.pend.j[bp.pend.n_jobs++];
    s.P = o.P; s.nb = o.nb; s.g = o.g0; s.gc = gc; s.g_grid = o.g_grid; s.perm = bp.perm; s.n = o.P.n; s.flags = flags; s.flag_mask = ADFP_F_BAND;
    return 0;
}
static void launch_pgrad_single(const DecodeBwdHArgs& a, int role, int blocks, hipStream_t st) {
    if (role == ROLE_HIGH) hipLaunchKernelGGL((k_decode_bwd_h<64, 1, ROLE_HIGH, false, false, 512, true>), dim3(blocks), dim3(512), 0, st, a);
    else if (role == ROLE_LOW) hipLaunchKernelGGL((k_decode_bwd_h<32, 1, ROLE_LOW, false, false, 512, true>), dim3(blocks), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((k_decode_bwd_h<32, 4, ROLE_COLOR, false, false, 512, true>), dim3(blocks), dim3(512), 0, st, a);
}
static int flush_pgrad(BinPlan& bp, BwdWorkspace& bw, hipStream_t st) {
    DecodeBwdH3Args& m = bp.pgrad;
    if (m.n == 0) return 0;
    if (m.n >= 2 && bw.pgrad_separate) {               // side by side, each into its own buffer (k_rays_grad adds them up)
        for (int k = 0; k < m.n; ++k) m.j[k].g_pts = bw.g_pts_dec + (size_t)k * bw.g_pts_stride;
        hipLaunchKernelGGL(k_decode_bwd_h_pgrad3, dim3(m.first[m.n]), dim3(512), 0, st, m);
        ADFP_CHECK_LAUNCH();
        bw.pgrad_used = m.n;
    } else {                                           // one after the other, accumulating into g_pts
        for (int k = 0; k < m.n; ++k) { launch_pgrad_single(m.j[k], m.role[k], m.first[k + 1] - m.first[k], st); ADFP_CHECK_LAUNCH(); }
    }
    m.n = 0;
    return 0;
}
static int flush_scatter(BinPlan& bp, hipStream_t st) {
    if (!bp.ok || bp.pend.n_jobs == 0) return 0;
    const int n = bp.pend.j[0].n;
    bp.pend.blocks_per_job = (n + ADFP_SCATTER_PPW * ADFP_SCATTER_NW - 1) / (ADFP_SCATTER_PPW * ADFP_SCATTER_NW);
    for (int k = bp.pend.n_jobs; k < ADFP_SCATTER_MAX_JOBS; ++k) bp.pend.j[k] = bp.pend.j[0];
    hipLaunchKernelGGL(k_scatter_sorted, dim3(bp.pend.blocks_per_job * bp.pend.n_jobs), dim3(64 * ADFP_SCATTER_NW), 0, st, bp.pend);
    ADFP_CHECK_LAUNCH();
    bp.pend.n_jobs = 0;
    return 0;
}

template <int CDIM, int NOUT, int ROLE>
static int run_decode_bwd_h(const DecodeBwdArgs& o, const void* t, const unsigned* masks, const float* act, int* status, const int* skip, int total,
                            const int* count_ptr, float* flat, BwdWorkspace& bw, BinPlan& bp, const unsigned char* flags, int options, hipStream_t st) {
    if (total == 0) return 0;
    // d/d c rows + k_scatter_sorted instead of the in-kernel scatter: for a grid on one of the two lattices the points were
    // sorted by (k_bin_keys); a third lattice keeps the in-kernel scatter
    const bool binned = bp.ok && o.g_grid && ((o.g0.X == bp.args.RX && o.g0.Y == bp.args.RY && o.g0.Z == bp.args.RZ) ||
                                              (o.g0.X == bp.args.CX && o.g0.Y == bp.args.CY && o.g0.Z == bp.args.CZ));
    if (!binned && o.g_grid && (long long)o.g0.X * o.g0.Y * o.g0.Z >= (1ll << 27)) return ADFP_E_UNSUPPORTED;     // scatter cache slot bits
    DecodeBwdHArgs a;
    a.P = o.P; a.nb = o.nb; a.g0 = o.g0; a.packed_t = (const unsigned*)t; a.list = o.list; a.count_ptr = o.count_ptr;
    a.g_raw = o.g_raw; a.att_g = o.att_g; a.masks = masks; a.g_grid = o.g_grid; a.stage = nullptr; a.status = status; a.gmax = bw.gmax; a.skip = skip;
    float* const gc_rows = binned ? bw.gc + (size_t)bp.pend.n_jobs * bw.gc_stride : nullptr;       // this grid's d/d c rows
    a.gc_out = gc_rows;
    a.g_pts = o.g_pts;
    constexpr int NW = ADFP_BWDH_NT / 64;
    if (o.g_pts) {                                      // position gradient only (backward_points' use_h): frozen network and grid
        if (flat || o.g_grid) return ADFP_E_ARG;
        a.chunk_lo = 0; a.chunk_hi = total;
        // deferred: the call's decoders run side by side in ONE launch (flush_pgrad at the end of backward_points)
        DecodeBwdH3Args& m = bp.pgrad;
        if (m.n >= ADFP_PGRAD_MAX_JOBS) return ADFP_E_UNSUPPORTED;
        if (m.n == 0) m.first[0] = 0;
        m.j[m.n] = a; m.role[m.n] = ROLE;
        m.first[m.n + 1] = m.first[m.n] + decode_grid((total + 31) / 32, 8, 1);
        ++m.n;
        return 0;
    }
    if constexpr (CDIM == 32) {
        // weight gradients inside the chain kernel (adfp_backward_fused.h): no staging rows, no k_outer_h.  Needs the grid
        // gradient on the sorted scatter (or not wanted): the in-kernel scatter's LDS structures do not fit beside the reduction copy.
        if (flat && !count_ptr && (binned || !o.g_grid) && !(options & ADFP_BWD_STAGED_WGRAD)) {
            DecodeBwdFArgs f;
            f.P = o.P; f.nb = o.nb; f.packed_t = (const unsigned*)t; f.g_raw = o.g_raw; f.masks = masks; f.act = act;
            f.gc_out = gc_rows; f.total = total; f.status = status; f.gmax = bw.gmax; f.skip = skip;
            f.partial = bw.partial; f.part_stride = bw.part_stride;
            // every workgroup OVERWRITES its slot of bw.partial (no 20 MB zero fill per network), and the reduction reads the slots in use
            const int ntiles = (total + 31) / 32;
            int nslot;
            // (role P stages a tile's positions through LDS and allows a 32-point tile to span five rays: fewer than 8 samples per
            // ray keep the one-wave kernel)
            if ((options & ADFP_BWD_FUSED_ONE_WAVE) || (o.P.mode == ADFP_PTS_RAYS && o.P.S < 8)) {
                const int nwg = (ntiles + 3) / 4;
                nslot = nwg < OUTER_NSLOT ? nwg : OUTER_NSLOT;
                hipLaunchKernelGGL((k_decode_bwd_fused<NOUT, ROLE>), dim3(nslot), dim3(256), 0, st, f);
                ADFP_CHECK_LAUNCH();
                hipLaunchKernelGGL(k_reduce_partials_scaled, dim3((DecLayout<CDIM, NOUT>::F_TOTAL + 31) / 32), dim3(256), 0, st, bw.partial, nslot, bw.part_stride,
                                   DecLayout<CDIM, NOUT>::F_TOTAL, flat, bw.gmax);
            } else {
                // role-split kernel (adfp_backward_roles.h): one 512-thread workgroup per CU, dealt to the three roles in proportion to
                // what a tile costs each; every role walks all tiles, so a small call still wants all three
                int cus = num_cu(); if (cus > OUTER_NSLOT) cus = OUTER_NSLOT;
                int g = 3 * ((ntiles + 7) / 8);
                nslot = g < 3 ? 3 : (g > cus ? cus : g);
                // (a -DADFP_TUNE_ROLE_SHARES build takes the split from the environment: how the shares were tuned, tools/build_ab_libs.sh)
                static const int share_p = role_share_env(0, ROLE_SHARE_P), share_h = role_share_env(1, ROLE_SHARE_H);
                int nP = (nslot * share_p + 128) / 256, nH = (nslot * share_h + 128) / 256;
                if (nP < 1) nP = 1;
                if (nH < 1) nH = 1;
                while (nP + nH > nslot - 1) { if (nP > nH) --nP; else --nH; }
                hipLaunchKernelGGL((k_decode_bwd_roles<NOUT, ROLE>), dim3(nslot), dim3(512), 0, st, f, nP, nH);
                ADFP_CHECK_LAUNCH();
                // a slot holds only the elements of its workgroup's role: each element is added up over its owners
                hipLaunchKernelGGL((k_reduce_partials_roles<CDIM, NOUT>), dim3((DecLayout<CDIM, NOUT>::F_TOTAL + 31) / 32), dim3(256), 0, st, bw.partial, nP, nH, nslot,
                                   bw.part_stride, flat, bw.gmax);
            }
            ADFP_CHECK_LAUNCH();
            return binned ? scatter_bins(bp, o, gc_rows, flags) : 0;
        }
    }
    if (!flat) {
        a.chunk_lo = 0; a.chunk_hi = total;
        if (o.g_grid && !binned) hipLaunchKernelGGL((k_decode_bwd_h<CDIM, NOUT, ROLE, false, true, ADFP_BWDH_NT>), dim3(decode_grid((total + 31) / 32, NW, 1)), dim3(ADFP_BWDH_NT), 0, st, a);
        else hipLaunchKernelGGL((k_decode_bwd_h<CDIM, NOUT, ROLE, false, false, 512>), dim3(decode_grid((total + 31) / 32, 8, 1)), dim3(512), 0, st, a);
        ADFP_CHECK_LAUNCH();
        return binned ? scatter_bins(bp, o, gc_rows, flags) : 0;
    }
    using ST = DecStage<CDIM>;
    OuterHArgs oa; decoder_jobs<CDIM, NOUT>(oa.o);
    oa.act = act; oa.nxm4 = ST::NXM / 4; oa.ngm4 = ST::NGM / 4; oa.g_dst4 = ST::SGH(0) / 4; oa.x_gap_at4 = 8; oa.x_gap4 = 24; oa.x_skip4 = 8; oa.P = o.P; oa.list = o.list; oa.masks = masks; oa.bm = (const float*)t;   // P_BM = word 0 of the T image
    oa.col_se = ST::SE; oa.col_sgp = ST::SGP(0); oa.status = status; oa.skip = skip; oa.overwrite = 0;
    a.stage = bw.stage;
    const int rows_cap = (int)((size_t)bw.stage_rows * AttStage::NCOLS / ST::NGM);
    int rc = outer_begin(bw, DecLayout<CDIM, NOUT>::F_TOTAL, st);
    if (rc) return rc;
    for (int lo = 0; lo < total; lo += rows_cap) {
        const int hi = lo + rows_cap < total ? lo + rows_cap : total;
        a.chunk_lo = lo; a.chunk_hi = hi;
        if (o.g_grid && !binned) hipLaunchKernelGGL((k_decode_bwd_h<CDIM, NOUT, ROLE, true, true, ADFP_BWDH_NT>), dim3(decode_grid((hi - lo + 31) / 32, NW, 1)), dim3(ADFP_BWDH_NT), 0, st, a);
        else hipLaunchKernelGGL((k_decode_bwd_h<CDIM, NOUT, ROLE, true, false, 512>), dim3(decode_grid((hi - lo + 31) / 32, 8, 1)), dim3(512), 0, st, a);
        ADFP_CHECK_LAUNCH();
        oa.o.stage = bw.stage; oa.o.count_ptr = count_ptr; oa.o.chunk_lo = lo; oa.o.chunk_hi = hi; oa.o.flat = flat;
        oa.o.partial = bw.partial; oa.o.part_stride = bw.part_stride;
        {   // block = a workgroup's whole share: computed here when the row count is known, by the kernel for a list (count on the device: br = 0)
            const int rows = hi - lo;
            int br = 0;
            if (!count_ptr) { br = (rows + OUTER_NSLOT - 1) / OUTER_NSLOT; br = ((br < 64 ? 64 : br) + OUTER_RT - 1) / OUTER_RT * OUTER_RT; }
            const int nblk = br ? (rows + br - 1) / br : (rows + OUTER_RT - 1) / OUTER_RT;
            oa.o.rows_per_wave = br;
            hipLaunchKernelGGL(k_outer_h, dim3(nblk < OUTER_NSLOT ? nblk : OUTER_NSLOT), dim3(512), 0, st, oa);
        }
        ADFP_CHECK_LAUNCH();
    }
    rc = outer_end_scaled(bw, DecLayout<CDIM, NOUT>::F_TOTAL, flat, st);
    if (rc) return rc;
    return binned ? scatter_bins(bp, o, gc_rows, flags) : 0;
}

template <int CDIM, int NOUT, int ROLE>
static int run_decode_bwd(DecodeBwdArgs a, int total, const int* count_ptr, float* flat, BwdWorkspace& bw, hipStream_t st) {
    return a.g_pts ? run_decode_bwd_p<CDIM, NOUT, ROLE, true>(a, total, count_ptr, flat, bw, st)
                   : run_decode_bwd_p<CDIM, NOUT, ROLE, false>(a, total, count_ptr, flat, bw, st);
}

// gradient outputs shared by the two backward entries
struct GradOut { float* grid_low; float* grid_high; float* grid_color; float* flat_low; float* flat_high; float* flat_color; float* flat_att; };

static int zero_grad_jobs(const adfp_scene* sc, const GradOut& g, int options, ZeroBatch& zb, hipStream_t st);
static int zero_grad_outputs(const adfp_scene* sc, const GradOut& g, int options, hipStream_t st) {
    ZeroBatch zb;
    int rc = zero_grad_jobs(sc, g, options, zb, st); if (rc) return rc;
    return (int)zb.flush(st);
}
static int zero_grad_jobs(const adfp_scene* sc, const GradOut& g, int options, ZeroBatch& zb, hipStream_t st) {
    const bool zg = !(options & ADFP_BWD_GRIDS_PREZEROED);
    struct { float* p; size_t n; } zs[7] = {
        {zg ? g.grid_low : nullptr, (size_t)sc->low.Z * sc->low.Y * sc->low.X * 32},
        {zg ? g.grid_high : nullptr, (size_t)sc->high.Z * sc->high.Y * sc->high.X * 32},
        {zg ? g.grid_color : nullptr, (size_t)sc->color.Z * sc->color.Y * sc->color.X * 32},
        {g.flat_low, (size_t)DecLayout<32, 1>::F_TOTAL}, {g.flat_high, (size_t)DecLayout<64, 1>::F_TOTAL},
        {g.flat_color, (size_t)DecLayout<32, 4>::F_TOTAL}, {g.flat_att, (size_t)AttLayout::F_TOTAL}};
    for (int k = 0; k < 7; ++k)
        if (zs[k].p) { hipError_t e = zb.add(zs[k].p, zs[k].n * 4, st); if (e != hipSuccess) return (int)e; }
    return 0;
}

// DF.forward backward over P points: bw.g_raw holds the cotangent of raw [P,4] (the attention pass reads .w and rewrites it
// with d/d(high+low)); g_pts (bw.g_pts, zeroed here) receives d/d position when pgrad.
static int backward_points(const adfp_scene* sc, int stage, const PtsDev& Pd, int P, const adfp_train_state& state, const float* g_weight,
                           const GradOut& go, bool pgrad, int options, BwdWorkspace& bw, hipStream_t st) {
    int rc;
    hipError_t e;
    // whatever path leaves this function: the main stream has waited for the side lane (a capture must see it joined)
    struct SideJoin { BwdWorkspace& bw; hipStream_t st;
                      hipError_t join() {
                          if (!bw.side_join_pending) return hipSuccess;
                          bw.side_join_pending = false;
                          hipError_t e_ = hipEventRecord(bw.side_ev[1], bw.side);            // after the lane's last launch
                          return e_ == hipSuccess ? hipStreamWaitEvent(st, bw.side_ev[1], 0) : e_;
                      }
                      ~SideJoin() { (void)join(); } } side_join{bw, st};
    struct CuReserve { int saved; CuReserve() : saved(t_cu_reserve) {} ~CuReserve() { t_cu_reserve = saved; } } cu_reserve;     // see num_cu()
    const bool fuse = stage != ADFP_STAGE_LOW;
    DecodeBwdArgs a;
    a.P = Pd; a.nb = make_norm(sc->bound); fill_bound(a.b, sc->bound);
    a.list = nullptr; a.count_ptr = nullptr; a.g_raw = bw.g_raw; a.att_g = nullptr; a.stage = nullptr; a.dbg_masks = nullptr;
    a.g_pts = pgrad ? bw.g_pts : nullptr;
    if (pgrad && !(bw.head_pending && bw.head_zeroes_g_pts)) { e = zero_async(bw.g_pts, (size_t)P * 12, st); if (e != hipSuccess) return (int)e; }
    // a decoder takes the f16-split backward when its T image and the forward's masks are there and -- if its weight gradients
    // are wanted -- the forward also left the layer inputs.  A position gradient comes from the f16-split kernels only on its own
    // (the Tracker: networks and grids frozen); together with weight or grid gradients (bundle adjustment) the exact kernels run.
    auto use_h = [&](const void* t, const unsigned* masks, const float* act, const float* flat, const float* grid = nullptr) {
        return t && masks && (!flat || act) && (!pgrad || (!flat && !grid));
    };
    // The grid gradients of the decoders that take the f16-split backward are scattered in spatial order (k_scatter_sorted): one
    // radix sort of the points by (coarsest such grid's cell, finest grid's cell inside it), shared by all of them.
    BinPlan bp; bp.ok = false; bp.pend.n_jobs = 0; bp.pgrad.n = 0;
    {
        const bool h_low = go.grid_low && use_h(sc->ht_low, state.masks_low, state.act_low, go.flat_low, go.grid_low);
        const bool h_high = fuse && go.grid_high && use_h(sc->ht_high, state.masks_high, state.act_high, go.flat_high, go.grid_high);
        const bool h_color = stage == ADFP_STAGE_COLOR && go.grid_color && use_h(sc->ht_color, state.masks_color, state.act_color, go.flat_color, go.grid_color);
        const adfp_grid* fine = nullptr; const adfp_grid* coarse = nullptr;
        auto vox = [](const adfp_grid& g) { return (long long)g.X * g.Y * g.Z; };
        const adfp_grid* cand[3] = {h_low ? &sc->low : nullptr, h_high ? &sc->high : nullptr, h_color ? &sc->color : nullptr};
        for (int k = 0; k < 3; ++k) {
            if (!cand[k]) continue;
            if (!fine || vox(*cand[k]) > vox(*fine)) fine = cand[k];
            if (!coarse || vox(*cand[k]) < vox(*coarse)) coarse = cand[k];
        }
        // ADFP_BWD_SCATTER_IN_KERNEL keeps every grid on the in-kernel write-combining scatter (what the tests compare the sorted
        // scatter against)
        const bool force_cache = (options & ADFP_BWD_SCATTER_IN_KERNEL) != 0;
        if (fine && !force_cache && coarse->X >= 2 && coarse->Y >= 2 && coarse->Z >= 2 && P > 0) {
            int bits = 0;
            while ((1 << bits) < coarse->X || (1 << bits) < coarse->Y || (1 << bits) < coarse->Z) ++bits;
            if (bits <= ADFP_BIN_MAXBITS) {
                BinArgs& b = bp.args;
                b.P = Pd; b.nb = a.nb; b.CX = coarse->X; b.CY = coarse->Y; b.CZ = coarse->Z; b.RX = fine->X; b.RY = fine->Y; b.RZ = fine->Z;
                b.key = bw.bin_key; b.val = bw.bin_val;
                const float* fold_parts = nullptr;            // the per-ray maxima still to be folded: by the sort's first launch when the
                int fold_n = 0;                               // head carries k_composite_bwd (they are not written yet), else by k_bin_keys
                if (bw.head_pending) {
                    rc = launch_backward_head(bw, &b, P, st); if (rc) return rc;
                    if (bw.gmax_pending > 0) { fold_parts = bw.gmax_parts; fold_n = bw.gmax_pending; }
                } else {
                    b.max_parts = bw.gmax_pending > 0 ? bw.gmax_parts : nullptr; b.max_n = bw.gmax_pending; b.max_out = bw.gmax;
                    hipLaunchKernelGGL(k_bin_keys, dim3((P + 255) / 256 + (b.max_parts ? 1 : 0)), dim3(256), 0, st, b);
                    ADFP_CHECK_LAUNCH();
                }
                bw.gmax_pending = 0;
                // stable LSD radix sort (adfp_sort.h), ping-pong between the two buffer pairs -- on the caller's side lane when there is
                // one: nine short launches nobody but k_scatter_sorted waits for, beside the backward kernels instead of in front of them
                const int* vfin = nullptr;
                hipStream_t sort_st = st;
                if (bw.side) {
                    // The gradient scale is folded HERE, on the main stream (one small launch): riding on the sort's first launch it would
                    // make the first backward kernel wait for the side lane -- two cross-stream hops of ~15 us each (measured: 36 us
                    // between k_backward_head's end and k_attention_bwd_h's start).
                    if (fold_parts) {
                        hipLaunchKernelGGL(k_max_reduce, dim3(1), dim3(1024), 0, st, fold_parts, fold_n, bw.gmax);
                        ADFP_CHECK_LAUNCH();
                        fold_parts = nullptr; fold_n = 0;
                    }
                    e = hipEventRecord(bw.side_ev[0], st);                       // the keys exist
                    if (e == hipSuccess) e = hipStreamWaitEvent(bw.side, bw.side_ev[0], 0);
                    if (e != hipSuccess) return (int)e;
                    sort_st = bw.side;
                    bw.side_join_pending = true;
                    t_cu_reserve = ADFP_SIDE_CU_RESERVE;          // the whole-CU kernels below leave the lane room (restored on return)
                }
                rc = radix_sort_pairs(bw.bin_key, bw.bin_val, bw.bin_key_sorted, bw.bin_perm, P, bin_key_bits(coarse->X, coarse->Y, coarse->Z), bw.sort_table, nullptr, &vfin, sort_st,
                                      fold_parts, fold_n, bw.gmax);
                if (rc) return rc;
                const int* vin = vfin;
                bp.perm = vin;                      // the pair the last pass wrote
                bp.ok = true;
            }
        }
    }

    if (bw.head_pending) { rc = launch_backward_head(bw, nullptr, P, st); if (rc) return rc; }      // no sorted scatter: zero fill + compositing backward
    if (bw.gmax_pending > 0) {                       // no sorted scatter in this call: the fold is a launch of its own
        hipLaunchKernelGGL(k_max_reduce, dim3(1), dim3(1024), 0, st, bw.gmax_parts, bw.gmax_pending, bw.gmax);
        ADFP_CHECK_LAUNCH();
        bw.gmax_pending = 0;
    }
    if (fuse) {
        const bool att_full_h = use_h(sc->ht_att, state.masks_att, state.act_att, go.flat_att);
        if (att_full_h) {
            // the attention backward on f16 MFMA (k_attention_bwd_h): masks, softmax weights and layer inputs from the forward
            AttBwdHArgs t;
            t.packed_t = (const unsigned*)sc->ht_att; t.list = state.list; t.count_ptr = state.counter; t.att_occ = state.att_occ; t.att_u = state.att_u;
            t.masks = state.masks_att; t.g_weight = g_weight; t.g_raw = bw.g_raw; t.att_g = bw.att_g; t.stage = bw.stage;
            t.status = sc->status; t.gmax = bw.gmax; t.skip = state.counter ? state.counter + 8 : nullptr;
            t.P = Pd; t.nt = make_norm(sc->tsdf_bnds); t.t = make_tsdf(sc->tsdf); t.g_pts = a.g_pts;
            OuterHArgs oh; attention_jobs(oh.o);
            oh.act = state.act_att; oh.nxm4 = 416 / 4; oh.ngm4 = 416 / 4; oh.g_dst4 = 416 / 4; oh.x_gap_at4 = 1 << 20; oh.x_gap4 = 0; oh.x_skip4 = 0; oh.P = PtsDev{}; oh.list = nullptr;
            oh.masks = nullptr; oh.bm = nullptr; oh.col_se = 0; oh.col_sgp = 0; oh.status = sc->status; oh.skip = t.skip;
            OuterArgs& oa = oh.o;
            const int rows_cap = bw.stage_rows * 2;               // the G piece is half a row
            // ONE chunk (the usual case: STG_ROWS_MAX): the weight-gradient workgroups OVERWRITE their slots and the reduction reads only
            // the slots in use -- no zero fill of the 256 x 134 KB partial sums (9.5 us + a launch per iteration)
            const bool one_chunk = P <= rows_cap;
            oh.overwrite = one_chunk ? 1 : 0;
            // one private gradient copy per workgroup; with one chunk the reduction reads the slots in use, so the launch may leave
            // compute units to the side lane (num_cu()); several chunks accumulate into all OUTER_NSLOT zero-filled slots
            const int att_slots = one_chunk ? (num_cu() < OUTER_NSLOT ? num_cu() : OUTER_NSLOT) : OUTER_NSLOT;
            if (go.flat_att && !one_chunk) { rc = outer_begin(bw, AttLayout::F_TOTAL, st); if (rc) return rc; }
            for (int lo = 0; lo < P; lo += rows_cap) {
                const int hi = lo + rows_cap < P ? lo + rows_cap : P;
                t.chunk_lo = lo; t.chunk_hi = hi;
                const int ntiles = (hi - lo + 31) / 32;
                if (go.flat_att) hipLaunchKernelGGL(k_attention_bwd_h<true>, dim3(decode_grid(ntiles, 8, 1)), dim3(512), 0, st, t);
                else if (pgrad) hipLaunchKernelGGL((k_attention_bwd_h<false, true>), dim3(decode_grid(ntiles, 8, 1)), dim3(512), 0, st, t);
                else hipLaunchKernelGGL(k_attention_bwd_h<false>, dim3(decode_grid(ntiles, 8, 1)), dim3(512), 0, st, t);
                ADFP_CHECK_LAUNCH();
                if (go.flat_att) {
                    oa.stage = bw.stage; oa.count_ptr = state.counter; oa.chunk_lo = lo; oa.chunk_hi = hi; oa.flat = go.flat_att;
                    oa.partial = bw.partial; oa.part_stride = bw.part_stride;
                    const int nblk = (hi - lo + OUTER_RT - 1) / OUTER_RT;     // list-based: the count is on the device, the kernel splits its rows evenly (even_block)
                    oa.rows_per_wave = 0;
                    hipLaunchKernelGGL(k_outer_h, dim3(nblk < att_slots ? nblk : att_slots), dim3(512), 0, st, oh);
                    ADFP_CHECK_LAUNCH();
                }
            }
            if (go.flat_att) {
                if (one_chunk) {
                    const int nblk = (P + OUTER_RT - 1) / OUTER_RT;
                    hipLaunchKernelGGL(k_reduce_partials_scaled, dim3((AttLayout::F_TOTAL + 31) / 32), dim3(256), 0, st, bw.partial, nblk < att_slots ? nblk : att_slots,
                                       bw.part_stride, AttLayout::F_TOTAL, go.flat_att, bw.gmax, state.counter, P, 0);
                    ADFP_CHECK_LAUNCH();
                } else { rc = outer_end_scaled(bw, AttLayout::F_TOTAL, go.flat_att, st); if (rc) return rc; }
            }
        } else {
        if (!sc->w_att) return ADFP_E_ARG;
        AttBwdArgs t;
        t.packed = sc->w_att; t.list = state.list; t.count_ptr = state.counter; t.att_occ = state.att_occ;
        t.att_u = state.att_u; t.g_weight = g_weight; t.g_raw = bw.g_raw; t.att_g = bw.att_g; t.stage = bw.stage;
        t.P = Pd; t.nt = make_norm(sc->tsdf_bnds); t.t = make_tsdf(sc->tsdf); t.g_pts = a.g_pts;
        t.gmax = nullptr; t.skip = state.counter ? state.counter + 8 : nullptr; t.dbg_masks = state.dbg_masks_att;
        OuterArgs oa; attention_jobs(oa);
        if (go.flat_att) { rc = outer_begin(bw, AttLayout::F_TOTAL, st); if (rc) return rc; }
        for (int lo = 0; lo < P; lo += bw.stage_rows) {
            const int hi = lo + bw.stage_rows < P ? lo + bw.stage_rows : P;
            t.chunk_lo = lo; t.chunk_hi = hi;
            const int ntiles = (hi - lo + 31) / 32;
            if (go.flat_att) {
                if (pgrad) hipLaunchKernelGGL((k_attention_bwd<true, true>), dim3(decode_grid(ntiles, 4, 1)), dim3(256), 0, st, t);
                else hipLaunchKernelGGL((k_attention_bwd<true, false>), dim3(decode_grid(ntiles, 4, 1)), dim3(256), 0, st, t);
                ADFP_CHECK_LAUNCH();
                rc = launch_outer(oa, bw, state.counter, lo, hi, go.flat_att, st);
                if (rc) return rc;
            } else {
                if (pgrad) hipLaunchKernelGGL((k_attention_bwd<false, true>), dim3(decode_grid(ntiles, 4, 1)), dim3(256), 0, st, t);
                else hipLaunchKernelGGL((k_attention_bwd<false, false>), dim3(decode_grid(ntiles, 4, 1)), dim3(256), 0, st, t);
                ADFP_CHECK_LAUNCH();
            }
        }
        if (go.flat_att) { rc = outer_end(bw, AttLayout::F_TOTAL, go.flat_att, st); if (rc) return rc; }
        }
        if (go.grid_high || go.flat_high || pgrad) {
            DecodeBwdArgs hgh = a;
            hgh.g0 = make_grid(sc->high); hgh.g1 = make_grid(sc->low); hgh.packed = sc->w_high;
            hgh.list = state.list; hgh.count_ptr = state.counter; hgh.att_g = bw.att_g; hgh.g_grid = go.grid_high; hgh.dbg_masks = state.dbg_masks_high;
            if (use_h(sc->ht_high, state.masks_high, state.act_high, go.flat_high, go.grid_high))
                rc = run_decode_bwd_h<64, 1, ROLE_HIGH>(hgh, sc->ht_high, state.masks_high, state.act_high, sc->status, state.counter ? state.counter + 8 : nullptr, P, state.counter, go.flat_high, bw, bp, state.flags, options, st);
            else rc = sc->w_high ? run_decode_bwd<64, 1, ROLE_HIGH>(hgh, P, state.counter, go.flat_high, bw, st) : ADFP_E_ARG;
            if (rc) return rc;
        }
    }
    if (go.grid_low || go.flat_low || pgrad) {
        DecodeBwdArgs lw = a;
        lw.g0 = make_grid(sc->low); lw.g1 = lw.g0; lw.packed = sc->w_low; lw.g_grid = go.grid_low; lw.dbg_masks = state.dbg_masks_low;
        if (use_h(sc->ht_low, state.masks_low, state.act_low, go.flat_low, go.grid_low))
            rc = run_decode_bwd_h<32, 1, ROLE_LOW>(lw, sc->ht_low, state.masks_low, state.act_low, sc->status, state.counter ? state.counter + 8 : nullptr, P, nullptr, go.flat_low, bw, bp, nullptr, options, st);
        else rc = sc->w_low ? run_decode_bwd<32, 1, ROLE_LOW>(lw, P, nullptr, go.flat_low, bw, st) : ADFP_E_ARG;
        if (rc) return rc;
    }
    if (stage == ADFP_STAGE_COLOR && (go.grid_color || go.flat_color || pgrad)) {
        DecodeBwdArgs cl = a;
        cl.g0 = make_grid(sc->color); cl.g1 = cl.g0; cl.packed = sc->w_color; cl.g_grid = go.grid_color; cl.dbg_masks = state.dbg_masks_color;
        if (use_h(sc->ht_color, state.masks_color, state.act_color, go.flat_color, go.grid_color))
            rc = run_decode_bwd_h<32, 4, ROLE_COLOR>(cl, sc->ht_color, state.masks_color, state.act_color, sc->status, state.counter ? state.counter + 8 : nullptr, P, nullptr, go.flat_color, bw, bp, nullptr, options, st);
        else rc = sc->w_color ? run_decode_bwd<32, 4, ROLE_COLOR>(cl, P, nullptr, go.flat_color, bw, st) : ADFP_E_ARG;
        if (rc) return rc;
    }
    rc = flush_pgrad(bp, bw, st); if (rc) return rc;
    e = side_join.join();                            // the sorted order is what k_scatter_sorted reads
    if (e != hipSuccess) return (int)e;
    return flush_scatter(bp, st);
}

static int check_backward_scene(const adfp_scene* sc, int stage) {
    if (!sc) return ADFP_E_ARG;
    if (stage < ADFP_STAGE_LOW || stage > ADFP_STAGE_COLOR) return ADFP_E_ARG;
    if (grid_too_big(sc->low) || grid_too_big(sc->high) || grid_too_big(sc->color)) return ADFP_E_UNSUPPORTED;
    if (!sc->low.data || (stage >= ADFP_STAGE_HIGH && (!sc->high.data || !sc->tsdf.data)) || (stage == ADFP_STAGE_COLOR && !sc->color.data)) return ADFP_E_ARG;
    // every network needs its exact image or its T image (backward_points picks)
    if (!(sc->w_low || sc->ht_low) || (stage >= ADFP_STAGE_HIGH && (!(sc->w_high || sc->ht_high) || !(sc->w_att || sc->ht_att))) ||
        (stage == ADFP_STAGE_COLOR && !(sc->w_color || sc->ht_color))) return ADFP_E_ARG;
    return 0;
}

extern "C" int adfp_render_backward(const adfp_scene* sc, const adfp_backward_args* r, void* stream) {
    if (!r) return ADFP_E_ARG;
    int rc = check_backward_scene(sc, r->stage); if (rc) return rc;
    if (!r->rays_o || !r->rays_d || !r->z_vals || !r->raw || !r->workspace || r->n_rays < 0 || r->S <= 0) return ADFP_E_ARG;
    if (r->S > 64 * CB_MAXC) return ADFP_E_UNSUPPORTED;
    const long long Pn = (long long)r->n_rays * r->S;
    if (Pn > 0x7fffffffll) return ADFP_E_UNSUPPORTED;
    BwdWorkspace bw = carve_bwd(r->workspace, Pn);
    if (r->workspace_bytes < bw.bytes) return ADFP_E_WORKSPACE;
    const bool fuse = r->stage != ADFP_STAGE_LOW;
    if (fuse && (!r->state.flags || !r->state.list || !r->state.counter || !r->state.att_occ || !r->state.att_u)) return ADFP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const GradOut go = {r->g_grid_low, r->g_grid_high, r->g_grid_color, r->g_flat_low, r->g_flat_high, r->g_flat_color, r->g_flat_att};
    if (r->n_rays == 0) return zero_grad_outputs(sc, go, r->options, st);    // every non-NULL output is zeroed, then accumulated into
    const int P = (int)Pn;
    // the zero fill and the compositing backward leave with backward_points' first launch (k_backward_head)
    {
        ZeroBatch zb;
        rc = zero_grad_jobs(sc, go, r->options, zb, st); if (rc) return rc;
        bw.head_zeroes_g_pts = false;
        if ((r->g_rays_o || r->g_rays_d) && zb.z.n + 1 < ZERO_MULTI) {  // the position gradients' accumulator and the decoders' own buffers
            hipError_t ze = zb.add(bw.g_pts, (size_t)P * 12, st);       // ride along (the Tracker); contiguous: g_pts, then g_pts_dec
            if (ze == hipSuccess) ze = zb.add(bw.g_pts_dec, ADFP_PGRAD_MAX_JOBS * bw.g_pts_stride * 4, st);
            if (ze != hipSuccess) return (int)ze;
            bw.head_zeroes_g_pts = true; bw.pgrad_separate = true;
        }
        bw.head_zero = zb.z; bw.head_zero_blocks = zb.blocks;
        bw.head_zero.n = zb.z.n;
        CompositeBwdArgs& c = bw.head_comp;
        c.raw = r->raw; c.z = r->z_vals; c.n_rays = r->n_rays; c.S = r->S; c.g_depth = r->g_depth; c.g_var = r->g_uncertainty; c.g_color = r->g_color;
        c.g_raw = bw.g_raw; c.keep = r->ray_keep; c.gmax = bw.gmax_parts; c.g_weight = r->g_weight; c.skip = r->state.counter ? r->state.counter + 8 : nullptr;
        bw.head_pending = true;
    }
    bw.gmax_pending = r->n_rays;                     // folded into bw.gmax by the sort's first launch (or k_max_reduce)
    if (r->side_stream) {                            // the sort's own lane (adfp_backward_args.side_stream)
        if (!r->side_events[0] || !r->side_events[1] || r->side_stream == stream) return ADFP_E_ARG;
        bw.side = (hipStream_t)r->side_stream;
        for (int k = 0; k < 2; ++k) bw.side_ev[k] = (hipEvent_t)r->side_events[k];
    }

    PtsDev Pd;
    Pd.mode = ADFP_PTS_RAYS; Pd.S = r->S; Pd.n = P; Pd.pts = nullptr; Pd.ro = r->rays_o; Pd.rd = r->rays_d; Pd.z = r->z_vals;
    const bool pgrad = r->g_rays_o || r->g_rays_d;
    rc = backward_points(sc, r->stage, Pd, P, r->state, r->g_weight, go, pgrad, r->options, bw, st);
    if (rc) return rc;
    if (pgrad) {
        RaysGradExtra ex; ex.n = bw.pgrad_used;
        for (int k = 0; k < 3; ++k) ex.p[k] = k < bw.pgrad_used ? bw.g_pts_dec + (size_t)k * bw.g_pts_stride : nullptr;
        hipLaunchKernelGGL(k_rays_grad, dim3((r->n_rays + 3) / 4), dim3(256), 0, st, bw.g_pts, r->z_vals, r->n_rays, r->S,
                           r->g_rays_o, r->g_rays_d, ex);
        ADFP_CHECK_LAUNCH();
    }
    return 0;
}

// cotangent of Renderer.eval_points' raw -> the workspace copy the point backward consumes; where the forward replaced the
// occupancy by 100 (point outside `bound`, Renderer.py:64) nothing flows back into the decoders
__global__ __launch_bounds__(256) void k_evalpts_bwd_prep(PtsDev P, const float* __restrict__ g_raw_in, float* __restrict__ g_raw, double b0, double b1,
                                                          double b2, double b3, double b4, double b5, int apply_bound, float* __restrict__ gmax,
                                                          const float* __restrict__ g_w, const int* __restrict__ skip) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    float mx = 0.f;
    const bool dead = skip && *skip;                 // the forward call was repaired by the f32 fallback: zero gradients (adfp.h)
    if (q < P.n) {
        if (g_w && !dead) mx = fabsf(g_w[q]);        // the attention weight's cotangent shares the gradient scale (see k_composite_bwd)
        f32x4 g = (g_raw_in && !dead) ? *(const f32x4*)(g_raw_in + 4ll * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (apply_bound) {
            double pt[3]; load_point(P, q, pt);
            const double b[6] = {b0, b1, b2, b3, b4, b5};
            if (!in_bound(pt, b)) g.w = 0.f;
        }
        *(f32x4*)(g_raw + 4ll * q) = g;
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(g.x), fabsf(g.y)), fmaxf(fabsf(g.z), fabsf(g.w))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) gmax[blockIdx.x * 4 + (threadIdx.x >> 6)] = mx;          // per wave; k_max_reduce folds them (grad_scale)
}

extern "C" int adfp_eval_points_backward(const adfp_scene* sc, const adfp_points* pts, const adfp_points_backward_args* r, void* stream) {
    if (!r || !pts) return ADFP_E_ARG;
    int rc = check_backward_scene(sc, r->stage); if (rc) return rc;
    if (!r->workspace) return ADFP_E_ARG;
    PtsDev Pd; rc = make_pts(pts, &Pd); if (rc) return rc;
    BwdWorkspace bw = carve_bwd(r->workspace, Pd.n);
    if (r->workspace_bytes < bw.bytes) return ADFP_E_WORKSPACE;
    const bool fuse = r->stage != ADFP_STAGE_LOW;
    if (fuse && (!r->state.flags || !r->state.list || !r->state.counter || !r->state.att_occ || !r->state.att_u)) return ADFP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const GradOut go = {r->g_grid_low, r->g_grid_high, r->g_grid_color, r->g_flat_low, r->g_flat_high, r->g_flat_color, r->g_flat_att};
    rc = zero_grad_outputs(sc, go, r->options, st); if (rc) return rc;
    if (Pd.n == 0) return 0;
    hipLaunchKernelGGL(k_evalpts_bwd_prep, dim3((Pd.n + 255) / 256), dim3(256), 0, st, Pd, r->g_raw, bw.g_raw, sc->bound[0][0], sc->bound[0][1],
                       sc->bound[1][0], sc->bound[1][1], sc->bound[2][0], sc->bound[2][1], (r->flags & ADFP_EVAL_APPLY_BOUND) ? 1 : 0, bw.gmax_parts, r->g_w, r->state.counter ? r->state.counter + 8 : nullptr);
    ADFP_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_max_reduce, dim3(1), dim3(1024), 0, st, bw.gmax_parts, ((Pd.n + 255) / 256) * 4, bw.gmax);
    ADFP_CHECK_LAUNCH();
    rc = backward_points(sc, r->stage, Pd, Pd.n, r->state, r->g_w, go, r->g_pts != nullptr, r->options, bw, st);
    if (rc) return rc;
    if (r->g_pts) {
        hipError_t e = hipMemcpyAsync(r->g_pts, bw.g_pts, (size_t)Pd.n * 12, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}
