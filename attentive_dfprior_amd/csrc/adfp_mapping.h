// adfp_mapping.h -- the Mapper's per-call bookkeeping around the render path, on the device
// (SURVEY.md section 8f rank 4):
//   k_frustum_depth / k_frustum_mask   frustum feature selection, src/Mapper.py:90-158
//       (numpy + cv2.remap on the host in the reference, once per grid per mapping call)
//   k_masked_adam                      Adam on the masked voxels of a dense grid, in place:
//       replaces `val_grad = val[mask]` + `val[mask] = val_grad` before AND after every iteration
//       (src/Mapper.py:330-361, :382-388, :476-482) and torch.optim.Adam on the compact copy.
#pragma once
#include "adfp_device.h"

struct FrustumArgs {
    int X, Y, Z;                 // grid points per axis (val_shape[2], [1], [0])
    double lo[3], hi[3];         // self.bound
    float w2c[16];               // np.linalg.inv(c2w), row-major
    double fx, fy, cx, cy;
    int H, W;
    const float* depth;          // [H,W] current depth image
    float cam[3];                // c2w[:3,3]
    float* sampled;              // [X*Y*Z] cv2.remap result per grid point (workspace)
    unsigned* dmax_ord;          // max of `sampled`, order-preserving uint
    unsigned char* mask;         // [Z][Y][X] (the order the grid tensor [1,C,Z,Y,X] is stored in)
};

// torch.linspace(start, end, steps) in float32 as ATen computes it: step = (end-start)/(steps-1),
// lower half start + step*i, upper half end - step*(steps-1-i).
ADFP_DEV float linspace_f32(float start, float end, int steps, int i) {
    if (steps == 1) return start;
    const float step = (end - start) / (float)(steps - 1);
    return i < steps / 2 ? fmaf(step, (float)i, start) : fmaf(-step, (float)(steps - 1 - i), end);
}

// grid point `idx` (x-major order of torch.meshgrid(X, Y, Z), Mapper.py:105-109) -> pixel coordinates and
// camera depth exactly as Mapper.py:111-124 computes them (f32 transform, f64 intrinsics, f32 uv)
ADFP_DEV void frustum_project(const FrustumArgs& a, long long idx, float p[3], float& u, float& v, double& zc) {
    const int iz = (int)(idx % a.Z), iy = (int)((idx / a.Z) % a.Y), ix = (int)(idx / ((long long)a.Z * a.Y));
    p[0] = linspace_f32((float)a.lo[0], (float)a.hi[0], a.X, ix);
    p[1] = linspace_f32((float)a.lo[1], (float)a.hi[1], a.Y, iy);
    p[2] = linspace_f32((float)a.lo[2], (float)a.hi[2], a.Z, iz);
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        c[k] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(a.w2c[4 * k], p[0]), __fmul_rn(a.w2c[4 * k + 1], p[1])),
                                   __fmul_rn(a.w2c[4 * k + 2], p[2])), a.w2c[4 * k + 3]);
    c[0] = -c[0];                                            // cam_cord[:, 0] *= -1
    const double x = (double)c[0], y = (double)c[1], z = (double)c[2];
    const double uu = a.fx * x + 0.0 * y + a.cx * z;         // uv = K @ cam_cord
    const double vv = 0.0 * x + a.fy * y + a.cy * z;
    zc = (0.0 * x + 0.0 * y + 1.0 * z) + 1e-5;               // z = uv[:, -1:] + 1e-5
    u = (float)(uu / zc); v = (float)(vv / zc);
}

// cv2.remap(depth, u, v, INTER_LINEAR), BORDER_CONSTANT 0, float image: OpenCV rounds the map to 1/32 pixel
// (INTER_BITS = 5, cvRound = round-half-even), takes its four weights from the 32x32 bilinear table and
// substitutes the border value for taps outside the image.
ADFP_DEV float remap_linear(const float* __restrict__ img, int H, int W, float u, float v) {
    const float su = rintf(u * 32.f), sv = rintf(v * 32.f);
    // saturate like cvRound -> int then saturate_cast<short> of the integer part
    const int iu = (int)fminf(fmaxf(su, -2147483648.f), 2147483520.f), iv = (int)fminf(fmaxf(sv, -2147483648.f), 2147483520.f);
    int sx = iu >> 5, sy = iv >> 5;
    sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
    sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
    const float fx = (float)(iu & 31) * (1.f / 32.f), fy = (float)(iv & 31) * (1.f / 32.f);
    const float w00 = (1.f - fy) * (1.f - fx), w01 = (1.f - fy) * fx, w10 = fy * (1.f - fx), w11 = fy * fx;
    if (sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0) return 0.f;
    const bool x0 = sx >= 0 && sx < W, x1 = sx + 1 >= 0 && sx + 1 < W, y0 = sy >= 0 && sy < H, y1 = sy + 1 >= 0 && sy + 1 < H;
    const float v00 = (x0 && y0) ? img[(long long)sy * W + sx] : 0.f;
    const float v01 = (x1 && y0) ? img[(long long)sy * W + sx + 1] : 0.f;
    const float v10 = (x0 && y1) ? img[(long long)(sy + 1) * W + sx] : 0.f;
    const float v11 = (x1 && y1) ? img[(long long)(sy + 1) * W + sx + 1] : 0.f;
    return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(v00, w00), __fmul_rn(v01, w01)), __fmul_rn(v10, w10)), __fmul_rn(v11, w11));
}

__device__ __forceinline__ unsigned f2ord_m(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f_m(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// pass 1: the remapped depth of every grid point and its maximum (Mapper.py:126-140: zero depths are
// replaced by np.max(depths) over ALL grid points before the depth test)
__global__ __launch_bounds__(256) void k_frustum_depth(FrustumArgs a) {
    const long long n = (long long)a.X * a.Y * a.Z;
    unsigned m = 0;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
        float p[3], u, v; double zc;
        frustum_project(a, idx, p, u, v, zc);
        const float d = remap_linear(a.depth, a.H, a.W, u, v);
        a.sampled[idx] = d;
        const unsigned o = f2ord_m(d);
        m = o > m ? o : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = __shfl_xor(m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(a.dmax_ord, m);
}

// pass 2: pixel test, depth test, ball around the camera centre (Mapper.py:133-153), written in the
// [Z][Y][X] order of the grid tensor (the permute(2,1,0) of Mapper.py:345)
__global__ __launch_bounds__(256) void k_frustum_mask(FrustumArgs a) {
    const long long n = (long long)a.X * a.Y * a.Z;
    const float dmax = ord2f_m(*a.dmax_ord);
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
        float p[3], u, v; double zc;
        frustum_project(a, idx, p, u, v, zc);
        float d = a.sampled[idx];
        if (d == 0.f) d = dmax;
        bool m = (u < (float)a.W) & (u > 0.f) & (v < (float)a.H) & (v > 0.f);
        m = m & (0.0 <= -zc) & (-zc <= (double)__fadd_rn(d, 0.5f));
        const float dx = p[0] - a.cam[0], dy = p[1] - a.cam[1], dz = p[2] - a.cam[2];
        const float dist = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        m = m | (dist < 0.5f * 0.5f);
        const int iz = (int)(idx % a.Z), iy = (int)((idx / a.Z) % a.Y), ix = (int)(idx / ((long long)a.Z * a.Y));
        a.mask[((long long)iz * a.Y + iy) * a.X + ix] = m ? 1 : 0;
    }
}

// Adam (torch.optim.Adam, amsgrad off, weight_decay 0) on the voxels whose mask is set, all channels of
// a channel-major grid [C][nvox]; state tensors have the grid's shape and start at zero.  One thread per
// 4 consecutive voxels of one channel.
struct AdamArgs {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    const unsigned char* mask;   // [nvox] or NULL (= every element)
    long long nvox; int C;
    float beta1, beta2, eps;
    float step_size;             // lr / (1 - beta1^t)
    float sqrt_bc2;              // sqrt(1 - beta2^t)
    const float* derived;        // device {step_size, sqrt_bc2} written by k_adam_prep (graph-capturable steps), or NULL
};
ADFP_DEV void adam_one(const AdamArgs& a, long long i, float step_size, float sqrt_bc2) {
    const float g = a.grad[i];
    const float m = __fadd_rn(__fmul_rn(a.exp_avg[i], a.beta1), __fmul_rn(g, 1.f - a.beta1));               // mul_(b1).add_(g, alpha=1-b1)
    const float v = __fadd_rn(__fmul_rn(a.exp_avg_sq[i], a.beta2), __fmul_rn(__fmul_rn(1.f - a.beta2, g), g)); // mul_(b2).addcmul_(g, g, value=1-b2)
    a.exp_avg[i] = m; a.exp_avg_sq[i] = v;
    const float denom = __fadd_rn(__fdiv_rn(sqrtf(v), sqrt_bc2), a.eps);
    a.param[i] = __fadd_rn(a.param[i], __fdiv_rn(__fmul_rn(-step_size, m), denom));                      // addcdiv_(m, denom, value=-step_size)
}
ADFP_DEV void masked_adam_block(const AdamArgs& a, long long block);
__global__ __launch_bounds__(256) void k_masked_adam(AdamArgs a) { masked_adam_block(a, blockIdx.x); }
// several parameter groups in ONE launch (a Mapper iteration steps five: three grids and two networks)
#define ADFP_ADAM_MULTI 8
struct AdamMultiArgs { AdamArgs g[ADFP_ADAM_MULTI]; unsigned first_block[ADFP_ADAM_MULTI + 1]; int n; };
ADFP_DEV void masked_adam_multi_block(const AdamMultiArgs& m, unsigned blk) {
    int j = 0;
    while (j + 1 < m.n && blk >= m.first_block[j + 1]) ++j;
    masked_adam_block(m.g[j], (long long)blk - m.first_block[j]);
}
__global__ __launch_bounds__(256) void k_masked_adam_multi(AdamMultiArgs m) { masked_adam_multi_block(m, blockIdx.x); }
ADFP_DEV void masked_adam_block(const AdamArgs& a, long long block) {
    const long long quads = (a.nvox + 3) >> 2;
    const long long t = block * 256 + threadIdx.x;
    if (t >= quads * a.C) return;
    const long long c = t / quads, v0 = (t - c * quads) << 2;
    const bool full = v0 + 4 <= a.nvox;
    unsigned mk = 0x01010101u;
    if (a.mask) {
        if (full && ((a.nvox & 3) == 0)) mk = *(const unsigned*)(a.mask + v0);
        else { mk = 0; for (int k = 0; k < 4 && v0 + k < a.nvox; ++k) mk |= (unsigned)(a.mask[v0 + k] != 0) << (8 * k); }
    } else if (!full) { mk = 0; for (int k = 0; k < 4 && v0 + k < a.nvox; ++k) mk |= 1u << (8 * k); }
    if (mk == 0) return;
    const float step_size = a.derived ? a.derived[0] : a.step_size, sqrt_bc2 = a.derived ? a.derived[1] : a.sqrt_bc2;
    if (sqrt_bc2 == 0.f) return;                  // k_adam_prep's "skip this iteration" (sqrt(1 - beta2^t) is never 0 otherwise)
    const long long base = c * a.nvox + v0;
#pragma unroll
    for (int k = 0; k < 4; ++k) if ((mk >> (8 * k)) & 0xffu) adam_one(a, base + k, step_size, sqrt_bc2);
}

// Adam on channels-last state (adfp_adam_grids_cl): a workgroup owns 64 consecutive voxels = 2 048 consecutive floats of every
// channels-last buffer (coalesced 16-byte accesses), steps the masked ones, zeroes the gradient it consumed, and writes the new
// parameters through to the reference-layout grid [32][nvox] via a padded LDS tile so that those stores are 256-byte runs per
// channel.  Arithmetic = adam_one, element for element what torch.optim.Adam does.
struct AdamClArgs {
    float* p_cl; float* p_cm; float* g_cl; float* m_cl; float* v_cl; const unsigned char* mask; long long nvox;
    float beta1, beta2, eps; const float* derived;
};
#define ADFP_ADAM_CL_MULTI 8
struct AdamClMultiArgs { AdamClArgs g[ADFP_ADAM_CL_MULTI]; unsigned first_block[ADFP_ADAM_CL_MULTI + 1]; int n; };
ADFP_DEV void adam_cl_multi_block(const AdamClMultiArgs& m, unsigned blk) {
    __shared__ float tile[32][65];
    __shared__ unsigned char s_mask[64];
    int j = 0;
    while (j + 1 < m.n && blk >= m.first_block[j + 1]) ++j;
    const AdamClArgs& a = m.g[j];
    const long long v0 = ((long long)blk - m.first_block[j]) * 64;
    const int nv = a.nvox - v0 < 64 ? (int)(a.nvox - v0) : 64;
    if (threadIdx.x < 64) s_mask[threadIdx.x] = ((int)threadIdx.x < nv && (!a.mask || a.mask[v0 + threadIdx.x])) ? 1 : 0;
    __syncthreads();
    const float step_size = a.derived[0], sqrt_bc2 = a.derived[1];
    const bool stepping = sqrt_bc2 != 0.f;                    // k_adam_prep's "skip this iteration"
    bool any = false;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e4 = threadIdx.x + 256 * k;                 // f32x4 index inside the block's 64 x 32 floats
        const int vl = e4 >> 3, c0 = (e4 & 7) * 4;
        if (vl >= nv) continue;
        const long long off = (v0 + vl) * 32 + c0;
        const f32x4 g = *(const f32x4*)(a.g_cl + off);
        *(f32x4*)(a.g_cl + off) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!stepping || !s_mask[vl]) continue;
        any = true;
        f32x4 p = *(const f32x4*)(a.p_cl + off), mo = *(const f32x4*)(a.m_cl + off), vo = *(const f32x4*)(a.v_cl + off);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float mm = __fadd_rn(__fmul_rn(mo[q], a.beta1), __fmul_rn(g[q], 1.f - a.beta1));
            const float vv = __fadd_rn(__fmul_rn(vo[q], a.beta2), __fmul_rn(__fmul_rn(1.f - a.beta2, g[q]), g[q]));
            const float denom = __fadd_rn(__fdiv_rn(sqrtf(vv), sqrt_bc2), a.eps);
            p[q] = __fadd_rn(p[q], __fdiv_rn(__fmul_rn(-step_size, mm), denom));
            mo[q] = mm; vo[q] = vv;
            tile[c0 + q][vl] = p[q];
        }
        *(f32x4*)(a.p_cl + off) = p; *(f32x4*)(a.m_cl + off) = mo; *(f32x4*)(a.v_cl + off) = vo;
    }
    if (!__syncthreads_or(any ? 1 : 0)) return;                // nothing stepped in this block: the reference-layout grid is untouched
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;    // 64 voxels x 4 channel rows per pass
    if (tx < nv && s_mask[tx])
#pragma unroll
        for (int c = ty; c < 32; c += 4) a.p_cm[(long long)c * a.nvox + v0 + tx] = tile[c][tx];
}
__global__ __launch_bounds__(256) void k_adam_cl_multi(AdamClMultiArgs m) { adam_cl_multi_block(m, blockIdx.x); }
// every parameter group of a Mapper iteration in ONE launch (adfp_adam_step): the channels-last grids' workgroups first, then the flat
// network buffers' (two launches of 5 + 18 us were 5 us more than one)
struct AdamStepArgs { AdamClMultiArgs cl; AdamMultiArgs fl; unsigned cl_blocks; };
__global__ __launch_bounds__(256) void k_adam_step(AdamStepArgs s) {
    if (blockIdx.x < s.cl_blocks) adam_cl_multi_block(s.cl, blockIdx.x);
    else masked_adam_multi_block(s.fl, blockIdx.x - s.cl_blocks);
}
