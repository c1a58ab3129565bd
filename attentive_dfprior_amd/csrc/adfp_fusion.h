// adfp_fusion.h -- TSDF integration of one RGB-D frame into the HBM-resident volume (what the reference does with an inline
// PyCUDA kernel, src/fusion.py:69-142, launched block by block from TSDFVolume.integrate, :226-251).
//
// Layout and mapping (MI355X-first, not the reference's one-thread-per-voxel grid of 1-D blocks):
//   * the three volumes (tsdf, weight, packed colour) stay in the reference's physical order [X][Y][Z], Z fastest -- that buffer IS
//     the render path's permuted [1,1,Z,Y,X] view, nothing is converted between fusion and rendering;
//   * a lane owns a QUAD of four z-consecutive voxels = one aligned 16-byte piece of each volume, so a wave moves 1 KB per
//     volume per instruction (dwordx4 loads / stores, fully coalesced) and three volumes cost 3 loads + 3 stores per quad instead
//     of 12 + 12 scalar accesses;
//   * phase 1 is arithmetic only: the four voxels' projections and the depth lookups (the depth image is a few hundred KB and
//     lives in L2).  A quad none of whose voxels is hit -- behind the camera, outside the image, no depth, or more than the
//     truncation margin behind the surface: most of a 200 M-voxel room -- never touches the volumes, and a wave whose 256
//     voxels are all missed retires after phase 1 (one ballot);
//   * phase 2 loads the three 16-byte pieces, blends the hit voxels in registers and stores the pieces back (the quad is owned
//     by one lane: no atomics, untouched voxels are rewritten with the value just read).
//
// Arithmetic is the reference kernel's, float32 operation by operation, because the volume it produces is an INPUT of the
// render path and must be the reference's volume: the voxel coordinates come from the reference's float index decomposition
// (src/fusion.py:92-94 -- beyond 2^24 voxels (float)index is rounded, which moves the first voxels of some x-slabs one slab
// early; reproduced on purpose, an integer decomposition would give a different volume), every product and sum is rounded
// separately, roundf is C's.  What is NOT reproduced: the reference's bound test `index > N` (:89), which lets thread N
// read and write one element past the volume.
// PARITY PINNED (since round 4) against the reference's own kernel: the CUDA C string of src/fusion.py:69-142 is compiled by hipcc
// from where it lies (oracle/build_ref_fusion.py -> oracle/_ref/, once as written and once with the compiler's default
// contraction, which is what nvcc's -fmad=true under PyCUDA does) and tests/test_gpu_fusion.py holds this kernel AND the numpy
// restatement oracle.tsdf_integrate_np to the as-written build bit for bit (tsdf, weight and packed-colour volumes), to the
// contracted build within one ulp of the camera-space depth over the truncation margin, and -- independently -- to a float64
// fusion of the same frames away from pixel-rounding ties.
#pragma once
#include "adfp_device.h"

struct FuseFrame {
    float* sdf; float* wsum; float* rgb;      // [X][Y][Z] volumes; rgb (packed b*65536 + g*256 + r) may be NULL
    int nx, ny, nz;
    float org[3];
    float cell;                               // voxel edge
    float K[9];                               // intrinsics, row-major 3x3
    float T[16];                              // camera-to-world, row-major 4x4 (OpenCV convention, get_tsdf.py:79-80)
    const float* rgb_im;                      // [H,W] packed colour
    const float* z_im;                        // [H,W] depth
    int rows, cols;
    float band;                               // truncation margin
    float w_obs;                              // weight of this observation
};

// what one voxel sees of the frame
struct VoxelHit { bool hit; int pix; float sd; };      // pixel offset into the images, truncated signed distance / band (<= 1)

// linear index -> the reference's float (x, y, z) lattice coordinates (src/fusion.py:92-94)
ADFP_DEV void lattice_of_index(int lin, int ny, int nz, float& fx, float& fy, float& fz) {
    const int slab = ny * nz;
    fx = floorf(((float)lin) / ((float)slab));
    const int in_slab = lin - ((int)fx) * slab;
    fy = floorf(((float)in_slab) / ((float)nz));
    fz = (float)(in_slab - ((int)fy) * nz);
}

ADFP_DEV VoxelHit probe_voxel(const FuseFrame& f, int lin) {
    VoxelHit r; r.hit = false; r.pix = 0; r.sd = 0.f;
    float gx, gy, gz;
    lattice_of_index(lin, f.ny, f.nz, gx, gy, gz);
    // world position relative to the camera centre, then into the camera frame: R^T (p - t), sums left to right
    const float rx = __fsub_rn(__fadd_rn(f.org[0], __fmul_rn(gx, f.cell)), f.T[3]);
    const float ry = __fsub_rn(__fadd_rn(f.org[1], __fmul_rn(gy, f.cell)), f.T[7]);
    const float rz = __fsub_rn(__fadd_rn(f.org[2], __fmul_rn(gz, f.cell)), f.T[11]);
    float cam[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        cam[k] = __fadd_rn(__fadd_rn(__fmul_rn(f.T[k], rx), __fmul_rn(f.T[4 + k], ry)), __fmul_rn(f.T[8 + k], rz));
    const int u = (int)roundf(__fadd_rn(__fmul_rn(f.K[0], __fdiv_rn(cam[0], cam[2])), f.K[2]));
    const int v = (int)roundf(__fadd_rn(__fmul_rn(f.K[4], __fdiv_rn(cam[1], cam[2])), f.K[5]));
    if (u < 0 || u >= f.cols || v < 0 || v >= f.rows || cam[2] < 0) return r;          // outside the view frustum
    r.pix = v * f.cols + u;
    const float z = f.z_im[r.pix];
    if (z == 0) return r;                                                              // no measurement at this pixel
    const float ahead = __fsub_rn(z, cam[2]);
    if (ahead < -f.band) return r;                                                     // further behind the surface than the band
    r.sd = fminf(1.0f, __fdiv_rn(ahead, f.band));
    r.hit = true;
    return r;
}

// running average of the packed colour, channel by channel (src/fusion.py:130-141)
ADFP_DEV float blend_packed_rgb(float have, float seen, float w_have, float w_obs, float w_next) {
    const float hb = floorf(have / (256 * 256)), hg = floorf((have - hb * 256 * 256) / 256), hr = have - hb * 256 * 256 - hg * 256;
    const float sb = floorf(seen / (256 * 256)), sg = floorf((seen - sb * 256 * 256) / 256), sr = seen - sb * 256 * 256 - sg * 256;
    const float b = fminf(roundf(__fdiv_rn(__fadd_rn(__fmul_rn(hb, w_have), __fmul_rn(w_obs, sb)), w_next)), 255.0f);
    const float g = fminf(roundf(__fdiv_rn(__fadd_rn(__fmul_rn(hg, w_have), __fmul_rn(w_obs, sg)), w_next)), 255.0f);
    const float r = fminf(roundf(__fdiv_rn(__fadd_rn(__fmul_rn(hr, w_have), __fmul_rn(w_obs, sr)), w_next)), 255.0f);
    return b * 256 * 256 + g * 256 + r;
}

// quad q covers linear indices [4 q, 4 q + 4); the host launches ceil(n / 4) lanes.  Aligned 16-byte accesses need the volumes'
// base addresses 16-byte aligned (checked on the host; every torch allocation is) -- otherwise the scalar tail path runs.
template <bool VEC>
__global__ __launch_bounds__(256) void k_fuse_frame(FuseFrame f) {
    const long long total = (long long)f.nx * f.ny * f.nz;
    const long long quad = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long first = quad * 4;
    if (first >= total) return;
    const int count = total - first < 4 ? (int)(total - first) : 4;
    VoxelHit hit[4];
    bool any = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hit[e].hit = false;
        if (e < count) hit[e] = probe_voxel(f, (int)(first + e));
        any |= hit[e].hit;
    }
    if (__ballot(any) == 0ull) return;                  // the whole wave missed the frame: nothing to read or write
    if (!any) return;
    float d[4], w[4], c[4];
    if (VEC && count == 4) {
        const f32x4 dv = *(const f32x4*)(f.sdf + first), wv = *(const f32x4*)(f.wsum + first);
        const f32x4 cv = f.rgb ? *(const f32x4*)(f.rgb + first) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) { d[e] = dv[e]; w[e] = wv[e]; c[e] = cv[e]; }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < count) { d[e] = f.sdf[first + e]; w[e] = f.wsum[first + e]; c[e] = f.rgb ? f.rgb[first + e] : 0.f; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (!hit[e].hit) continue;
        const float w_next = __fadd_rn(w[e], f.w_obs);
        d[e] = __fdiv_rn(__fadd_rn(__fmul_rn(d[e], w[e]), __fmul_rn(f.w_obs, hit[e].sd)), w_next);     // weighted running mean
        if (f.rgb) c[e] = blend_packed_rgb(c[e], f.rgb_im[hit[e].pix], w[e], f.w_obs, w_next);
        w[e] = w_next;
    }
    if (VEC && count == 4) {
        *(f32x4*)(f.sdf + first) = f32x4{d[0], d[1], d[2], d[3]};
        *(f32x4*)(f.wsum + first) = f32x4{w[0], w[1], w[2], w[3]};
        if (f.rgb) *(f32x4*)(f.rgb + first) = f32x4{c[0], c[1], c[2], c[3]};
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < count && hit[e].hit) { f.sdf[first + e] = d[e]; f.wsum[first + e] = w[e]; if (f.rgb) f.rgb[first + e] = c[e]; }
    }
}
