// adfp_fusion.h -- TSDF integration of one RGB-D frame (reference: the inline CUDA kernel of
// src/fusion.py:69-142, launched from TSDFVolume.integrate, :226-251).  One thread per voxel; the
// volume is HBM-resident in its physical [X][Y][Z] order (Z fastest), i.e. exactly the buffer the
// render path reads as the permuted view [1,1,Z,Y,X] -- no host round trip between fusion and use.
//
// Arithmetic follows the reference's float32 expressions one by one, INCLUDING its voxel-index
// decomposition through float division (src/fusion.py:92-94): for volumes beyond 2^24 voxels
// (float)voxel_idx is rounded, which puts the first few voxels of some x-slabs one slab early.  It
// is reproduced here on purpose (results identical to the reference); the off-by-one bound test
// `voxel_idx > N` (:89), which lets thread N touch memory past the volume, is not.
//
// Rounding: every product and sum below is rounded separately (as the numpy restatement in oracle/ does).
// The reference's kernel string is compiled by nvcc, whose default -fmad=true may fuse some of these into
// fmas; which ones is a property of that compiler run, cannot be observed here (no pycuda, no CUDA device),
// and moves a result by at most one ulp before the pixel rounding -- part of why this entry is parity-unpinned.
#pragma once
#include "adfp_device.h"

struct FusionArgs {
    float* tsdf; float* weight; float* color;
    int dx, dy, dz;
    float origin[3];
    float voxel;
    float intr[9];      // row-major 3x3
    float pose[16];     // row-major 4x4 camera-to-world
    const float* color_im;   // [H,W] packed b*65536 + g*256 + r
    const float* depth_im;   // [H,W]
    int im_h, im_w;
    float trunc, obs_w;
};

__global__ __launch_bounds__(256) void k_tsdf_integrate(FusionArgs a) {
    const long long n = (long long)a.dx * a.dy * a.dz;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int voxel_idx = (int)idx;
    // voxel grid coordinates, through float like the reference
    const float voxel_x = floorf(((float)voxel_idx) / ((float)(a.dy * a.dz)));
    const float voxel_y = floorf(((float)(voxel_idx - ((int)voxel_x) * a.dy * a.dz)) / ((float)a.dz));
    const float voxel_z = (float)(voxel_idx - ((int)voxel_x) * a.dy * a.dz - ((int)voxel_y) * a.dz);
    // world, then camera coordinates (R^T (p - t)); separate mul/add roundings, no fma contraction
    const float pt_x = __fadd_rn(a.origin[0], __fmul_rn(voxel_x, a.voxel));
    const float pt_y = __fadd_rn(a.origin[1], __fmul_rn(voxel_y, a.voxel));
    const float pt_z = __fadd_rn(a.origin[2], __fmul_rn(voxel_z, a.voxel));
    const float tx = __fsub_rn(pt_x, a.pose[3]), ty = __fsub_rn(pt_y, a.pose[7]), tz = __fsub_rn(pt_z, a.pose[11]);
    const float cam_x = __fadd_rn(__fadd_rn(__fmul_rn(a.pose[0], tx), __fmul_rn(a.pose[4], ty)), __fmul_rn(a.pose[8], tz));
    const float cam_y = __fadd_rn(__fadd_rn(__fmul_rn(a.pose[1], tx), __fmul_rn(a.pose[5], ty)), __fmul_rn(a.pose[9], tz));
    const float cam_z = __fadd_rn(__fadd_rn(__fmul_rn(a.pose[2], tx), __fmul_rn(a.pose[6], ty)), __fmul_rn(a.pose[10], tz));
    const int pixel_x = (int)roundf(__fadd_rn(__fmul_rn(a.intr[0], __fdiv_rn(cam_x, cam_z)), a.intr[2]));
    const int pixel_y = (int)roundf(__fadd_rn(__fmul_rn(a.intr[4], __fdiv_rn(cam_y, cam_z)), a.intr[5]));
    if (pixel_x < 0 || pixel_x >= a.im_w || pixel_y < 0 || pixel_y >= a.im_h || cam_z < 0) return;
    const float depth_value = a.depth_im[pixel_y * a.im_w + pixel_x];
    if (depth_value == 0) return;
    const float depth_diff = __fsub_rn(depth_value, cam_z);
    if (depth_diff < -a.trunc) return;
    const float dist = fminf(1.0f, __fdiv_rn(depth_diff, a.trunc));
    const float w_old = a.weight[idx];
    const float w_new = __fadd_rn(w_old, a.obs_w);
    a.weight[idx] = w_new;
    a.tsdf[idx] = __fdiv_rn(__fadd_rn(__fmul_rn(a.tsdf[idx], w_old), __fmul_rn(a.obs_w, dist)), w_new);
    if (!a.color) return;
    const float old_color = a.color[idx];
    const float old_b = floorf(old_color / (256 * 256));
    const float old_g = floorf((old_color - old_b * 256 * 256) / 256);
    const float old_r = old_color - old_b * 256 * 256 - old_g * 256;
    const float new_color = a.color_im[pixel_y * a.im_w + pixel_x];
    float new_b = floorf(new_color / (256 * 256));
    float new_g = floorf((new_color - new_b * 256 * 256) / 256);
    float new_r = new_color - new_b * 256 * 256 - new_g * 256;
    new_b = fminf(roundf(__fdiv_rn(__fadd_rn(__fmul_rn(old_b, w_old), __fmul_rn(a.obs_w, new_b)), w_new)), 255.0f);
    new_g = fminf(roundf(__fdiv_rn(__fadd_rn(__fmul_rn(old_g, w_old), __fmul_rn(a.obs_w, new_g)), w_new)), 255.0f);
    new_r = fminf(roundf(__fdiv_rn(__fadd_rn(__fmul_rn(old_r, w_old), __fmul_rn(a.obs_w, new_r)), w_new)), 255.0f);
    a.color[idx] = new_b * 256 * 256 + new_g * 256 + new_r;
}
