// adfp_device.h -- device-side helpers shared by the gfx950 kernels of libadfp.so.
// CDNA4 only (wave64, MFMA f32 32x32x2); no portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "adfp.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ADFP_DEV __device__ __forceinline__

// flag bits written by the TSDF stage
#define ADFP_F_INBOUND 1u   // strictly inside Renderer.bound          (Renderer.py:51-54)
#define ADFP_F_BAND    2u   // -1+1e-4 < tsdf < 1-1e-4                 (decoder.py:329)

// ------------------------------------------------------------------------------------
// kernel-argument PODs (host fills them from the C-ABI structs)
// ------------------------------------------------------------------------------------
struct NormDev {          // p_n = ((p - lo) / (hi - lo)) * 2 - 1     (common.py:275-290)
    double lo[3];
    double inv[3];        // 1 / (hi - lo)
};

struct GridDev { const float* data; int Z, Y, X; float fZ1, fY1, fX1; };     // f*1 = (float)(dim - 1), from the host: stays in SGPRs
struct TsdfDev { const float* data; int Z, Y, X; long long sZ, sY, sX; const float* cb; };     // cb: corner-block copy (adfp_relayout_tsdf) or NULL

struct PtsDev {
    int mode;
    int S;
    int n;                // P  (< 2^31)
    const void* pts;
    const float* ro;
    const float* rd;
    const double* z;
};

// ------------------------------------------------------------------------------------
// points
// ------------------------------------------------------------------------------------
// pts = rays_o + rays_d * z_vals, f32*f64 -> f64 product then f64 add (Renderer.py:223);
// written with explicit _rn ops so the compiler cannot contract it into an fma.
ADFP_DEV void load_point(const PtsDev& P, int q, double p[3]) {
    if (P.mode == ADFP_PTS_RAYS) {
        const int r = (int)((unsigned)q / (unsigned)P.S);
        const double z = P.z[q];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            p[k] = __dadd_rn((double)P.ro[3 * r + k], __dmul_rn((double)P.rd[3 * r + k], z));
    } else if (P.mode == ADFP_PTS_F64) {
        const double* s = (const double*)P.pts + 3ll * q;
        p[0] = s[0]; p[1] = s[1]; p[2] = s[2];
    } else {
        const float* s = (const float*)P.pts + 3ll * q;
        p[0] = (double)s[0]; p[1] = (double)s[1]; p[2] = (double)s[2];
    }
}

// strict in-bound test in f64 (Renderer.py:51-54)
ADFP_DEV bool in_bound(const double p[3], const double b[6]) {
    return (p[0] < b[1]) & (p[0] > b[0]) & (p[1] < b[3]) & (p[1] > b[2]) & (p[2] < b[5]) & (p[2] > b[4]);
}

// normalise in f64, then .float() (common.py:275-290, decoder.py:171)
ADFP_DEV void normalize3(const NormDev& nb, const double p[3], float pn[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) pn[k] = (float)(((p[k] - nb.lo[k]) * nb.inv[k]) * 2.0 - 1.0);
}

// ------------------------------------------------------------------------------------
// trilinear lookup = F.grid_sample(5-D, 'bilinear', padding_mode='border', align_corners=True)
// (decoder.py:168-175).  ATen: unnormalise ((x+1)/2)*(size-1), clip to [0,size-1], floor.
// ------------------------------------------------------------------------------------
ADFP_DEV void tri_axis(float pn, int size, int& i0, int& i1, float& w0, float& w1) {
    float c = ((pn + 1.f) / 2.f) * (float)(size - 1);
    c = fminf(fmaxf(c, 0.f), (float)(size - 1));
    const float f = floorf(c);
    i0 = (int)f;
    w1 = c - f;                      // ix - ix_tnw
    w0 = (f + 1.f) - c;              // ix_bse - ix
    i1 = i0 + 1;
    if (i1 > size - 1) { i1 = size - 1; w1 = 0.f; }   // out-of-range corner contributes zero
}

// one scalar volume (the TSDF), arbitrary element strides.  The reference's volume is a permuted
// view whose fastest dimension is Z (get_tsdf.py:95-97): then the two z-corners of each (y,x)
// column are adjacent floats and are fetched with ONE 8-byte load (4 loads per point, not 8) --
// the stage is bound by the number of distinct cache lines the address unit walks per wave.
typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));
ADFP_DEV float trilerp_scalar(const TsdfDev& t, const float pn[3]) {
    int x0, x1, y0, y1, z0, z1; float wx0, wx1, wy0, wy1, wz0, wz1;
    tri_axis(pn[0], t.X, x0, x1, wx0, wx1);
    tri_axis(pn[1], t.Y, y0, y1, wy0, wy1);
    tri_axis(pn[2], t.Z, z0, z1, wz0, wz1);
    const float* d = t.data;
    const long long ox0 = x0 * t.sX, ox1 = x1 * t.sX, oy0 = y0 * t.sY, oy1 = y1 * t.sY;
    float v000, v001, v010, v011, v100, v101, v110, v111;
    if (t.sZ == 1 && t.Z >= 2) {
        const int zb = z0 < t.Z - 1 ? z0 : t.Z - 2;       // pair (zb, zb+1) always inside the volume
        const bool lo = z0 == zb;                          // false only when z0 = Z-1 (then z1 = z0, weight 0)
        const f32x2_u p00 = *(const f32x2_u*)(d + zb + oy0 + ox0);
        const f32x2_u p01 = *(const f32x2_u*)(d + zb + oy0 + ox1);
        const f32x2_u p10 = *(const f32x2_u*)(d + zb + oy1 + ox0);
        const f32x2_u p11 = *(const f32x2_u*)(d + zb + oy1 + ox1);
        v000 = lo ? p00.x : p00.y; v100 = p00.y;
        v001 = lo ? p01.x : p01.y; v101 = p01.y;
        v010 = lo ? p10.x : p10.y; v110 = p10.y;
        v011 = lo ? p11.x : p11.y; v111 = p11.y;
    } else {
        const long long oz0 = z0 * t.sZ, oz1 = z1 * t.sZ;
        v000 = d[oz0 + oy0 + ox0]; v001 = d[oz0 + oy0 + ox1];
        v010 = d[oz0 + oy1 + ox0]; v011 = d[oz0 + oy1 + ox1];
        v100 = d[oz1 + oy0 + ox0]; v101 = d[oz1 + oy0 + ox1];
        v110 = d[oz1 + oy1 + ox0]; v111 = d[oz1 + oy1 + ox1];
    }
    float o = v000 * ((wx0 * wy0) * wz0);          // tnw, tne, tsw, tse, bnw, bne, bsw, bse
    o = fmaf(v001, (wx1 * wy0) * wz0, o);
    o = fmaf(v010, (wx0 * wy1) * wz0, o);
    o = fmaf(v011, (wx1 * wy1) * wz0, o);
    o = fmaf(v100, (wx0 * wy0) * wz1, o);
    o = fmaf(v101, (wx1 * wy0) * wz1, o);
    o = fmaf(v110, (wx0 * wy1) * wz1, o);
    o = fmaf(v111, (wx1 * wy1) * wz1, o);
    return o;
}

// The same lookup split in two so that a caller can put the loads of SEVERAL points in flight before
// it consumes any of them (unit z stride only): prepare = indices, weights, 4 addresses; finish = blend.
struct TriPair {
    const f32x2_u* a00; const f32x2_u* a01; const f32x2_u* a10; const f32x2_u* a11;   // (y0,x0) (y0,x1) (y1,x0) (y1,x1)
    float wx0, wx1, wy0, wy1, wz0, wz1;
    bool lo;
};
ADFP_DEV void trilerp_pair_prepare(const TsdfDev& t, const float pn[3], TriPair& r) {
    int x0, x1, y0, y1, z0, z1;
    tri_axis(pn[0], t.X, x0, x1, r.wx0, r.wx1);
    tri_axis(pn[1], t.Y, y0, y1, r.wy0, r.wy1);
    tri_axis(pn[2], t.Z, z0, z1, r.wz0, r.wz1);
    const long long ox0 = x0 * t.sX, ox1 = x1 * t.sX, oy0 = y0 * t.sY, oy1 = y1 * t.sY;
    const int zb = z0 < t.Z - 1 ? z0 : t.Z - 2;       // pair (zb, zb+1) always inside the volume
    r.lo = z0 == zb;                                   // false only when z0 = Z-1 (then z1 = z0, weight 0)
    const float* d = t.data + zb;
    r.a00 = (const f32x2_u*)(d + oy0 + ox0); r.a01 = (const f32x2_u*)(d + oy0 + ox1);
    r.a10 = (const f32x2_u*)(d + oy1 + ox0); r.a11 = (const f32x2_u*)(d + oy1 + ox1);
}
ADFP_DEV float trilerp_pair_finish(const TriPair& r, f32x2_u p00, f32x2_u p01, f32x2_u p10, f32x2_u p11) {
    const float v000 = r.lo ? p00.x : p00.y, v100 = p00.y;
    const float v001 = r.lo ? p01.x : p01.y, v101 = p01.y;
    const float v010 = r.lo ? p10.x : p10.y, v110 = p10.y;
    const float v011 = r.lo ? p11.x : p11.y, v111 = p11.y;
    float o = v000 * ((r.wx0 * r.wy0) * r.wz0);          // same order as trilerp_scalar
    o = fmaf(v001, (r.wx1 * r.wy0) * r.wz0, o);
    o = fmaf(v010, (r.wx0 * r.wy1) * r.wz0, o);
    o = fmaf(v011, (r.wx1 * r.wy1) * r.wz0, o);
    o = fmaf(v100, (r.wx0 * r.wy0) * r.wz1, o);
    o = fmaf(v101, (r.wx1 * r.wy0) * r.wz1, o);
    o = fmaf(v110, (r.wx0 * r.wy1) * r.wz1, o);
    o = fmaf(v111, (r.wx1 * r.wy1) * r.wz1, o);
    return o;
}

// The same lookup out of the CORNER-BLOCK copy of the volume (adfp_relayout_tsdf): block (x0, y0, z0) holds the eight values
// v(min(x0 + dx, X-1), min(y0 + dy, Y-1), min(z0 + dz, Z-1)) at k = dx + 2 dy + 4 dz -- exactly the eight operands of the blend
// above, in its order -- as ONE aligned 32-byte piece, [X][Y][Z][8] with z fastest like the reference's volume.  A lookup is two
// adjacent 16-byte loads in one 64-byte sector wherever the point lies; the plain volume needs four 8-byte column pieces in
// four sectors (two of them Y Z 4 bytes apart) unless neighbouring lanes share them.  Same values, same order of operations:
// bit-identical results.  8 x the memory (config 5: 34 GB of 288), built once per volume.
struct TriBlock { const f32x4* a; float wx0, wx1, wy0, wy1, wz0, wz1; };
ADFP_DEV void trilerp_block_prepare(const TsdfDev& t, const float pn[3], TriBlock& r) {
    int x0, x1, y0, y1, z0, z1;
    tri_axis(pn[0], t.X, x0, x1, r.wx0, r.wx1);
    tri_axis(pn[1], t.Y, y0, y1, r.wy0, r.wy1);
    tri_axis(pn[2], t.Z, z0, z1, r.wz0, r.wz1);
    r.a = (const f32x4*)(t.cb + (((long long)x0 * t.Y + y0) * t.Z + z0) * 8);
}
ADFP_DEV float trilerp_block_finish(const TriBlock& r, f32x4 lo, f32x4 hi) {
    float o = lo.x * ((r.wx0 * r.wy0) * r.wz0);          // same order as trilerp_scalar
    o = fmaf(lo.y, (r.wx1 * r.wy0) * r.wz0, o);
    o = fmaf(lo.z, (r.wx0 * r.wy1) * r.wz0, o);
    o = fmaf(lo.w, (r.wx1 * r.wy1) * r.wz0, o);
    o = fmaf(hi.x, (r.wx0 * r.wy0) * r.wz1, o);
    o = fmaf(hi.y, (r.wx1 * r.wy0) * r.wz1, o);
    o = fmaf(hi.z, (r.wx0 * r.wy1) * r.wz1, o);
    o = fmaf(hi.w, (r.wx1 * r.wy1) * r.wz1, o);
    return o;
}

// 16 of the 32 channels of a channels-last feature voxel grid: lane-half `h` takes the channels
// kmapH(r,h) = 8q+4h .. 8q+4h+3 (q = 0..3), i.e. four 16-B pieces of each 128-B voxel line; the
// two halves of a lane pair cover the whole line (c[r] <-> channel kmapH(r, h)).
// tri_axis with (float)(size - 1) handed in: the int -> float conversion of a kernel argument is loop invariant, gets hoisted into
// a VGPR and -- in a kernel at its register budget -- spilled; as a float argument it is an SGPR operand
ADFP_DEV void tri_axis_f(float pn, int size, float fsize1, int& i0, int& i1, float& w0, float& w1) {
    float c = ((pn + 1.f) / 2.f) * fsize1;
    c = fminf(fmaxf(c, 0.f), fsize1);
    const float f = floorf(c);
    i0 = (int)f;
    w1 = c - f;
    w0 = (f + 1.f) - c;
    i1 = i0 + 1;
    if (i1 > size - 1) { i1 = size - 1; w1 = 0.f; }
}
ADFP_DEV void gather16(const GridDev& g, const float pn[3], int h, float* __restrict__ c) {
    int xi[2], yi[2], zi[2]; float wx[2], wy[2], wz[2];
    tri_axis_f(pn[0], g.X, g.fX1, xi[0], xi[1], wx[0], wx[1]);
    tri_axis_f(pn[1], g.Y, g.fY1, yi[0], yi[1], wy[0], wy[1]);
    tri_axis_f(pn[2], g.Z, g.fZ1, zi[0], zi[1], wz[0], wz[1]);
#pragma unroll
    for (int k = 0; k < 16; ++k) c[k] = 0.f;
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const float w = (wx[dx] * wy[dy]) * wz[dz];
                // byte offset of the voxel line in 32 bits (the host entries refuse grids beyond 2^31 bytes): scalar base +
                // 32-bit vector offset addressing instead of a 64-bit multiply-add chain per corner
                const unsigned off = (((unsigned)zi[dz] * (unsigned)g.Y + (unsigned)yi[dy]) * (unsigned)g.X + (unsigned)xi[dx]) * 128u + 16u * h;
                const f32x4* src = (const f32x4*)((const char*)g.data + off);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const f32x4 t = src[2 * v];
                    c[4 * v + 0] = fmaf(t.x, w, c[4 * v + 0]);
                    c[4 * v + 1] = fmaf(t.y, w, c[4 * v + 1]);
                    c[4 * v + 2] = fmaf(t.z, w, c[4 * v + 2]);
                    c[4 * v + 3] = fmaf(t.w, w, c[4 * v + 3]);
                }
            }
}

// ------------------------------------------------------------------------------------
// sin for Fourier features with |x| up to ~1e3 rad (B ~ N(0,25^2), decoder.py:21-22).
// The argument is reduced EXACTLY first, in turns: t = x/(2pi) - rint(x/(2pi)) by two fmas with a
// 2-constant split of 1/(2pi) (the product is exact inside the fma), and only the reduced turn
// fraction in [-0.5, 0.5] goes to the hardware v_sin_f32 (which computes sin(2 pi t)).
// Measured on MI355X over |x| < 3000 (tools/micro/sin_variants.hip): max abs error 3.2e-7, against
// 6.9e-8 for a full software sin/cos-polynomial version that costs 15 more VALU issue slots per
// feature; v_sin_f32 on the UNREDUCED argument would be off by ~1e-4.  VALU instructions do not
// hide behind MFMAs on this machine, so those slots are wall time (279 sines per sample).
// ------------------------------------------------------------------------------------
ADFP_DEV float adfp_turns(float x) {
    // 1/(2 pi) = C_HI + C_LO; x*C_HI - k is exact inside the fma (k = the nearest integer of it)
    // k = the integer nearest to x * C_HI by the magic-number addition (|x * C_HI| < 2^22; two full-rate instructions where
    // v_mul + v_rndne cost a full-rate and a quarter-rate one): (v + 1.5 * 2^23) - 1.5 * 2^23 rounds v to an integer, ties to even
    const float k = fmaf(x, 0.15915494f, 12582912.0f) - 12582912.0f;
    float t = fmaf(x, 0.15915494f, -k);
    return fmaf(x, 6.4206382432985265e-09f, t);
}
#ifdef ADFP_SIN_POLY
ADFP_DEV float adfp_sinf(float x) {
    const float k = rintf(x * 0.636619772f);
    float r = fmaf(k, -1.57079601e+00f, x);
    r = fmaf(k, -3.13916473e-07f, r);
    r = fmaf(k, -5.39030253e-15f, r);
    const int n = (int)k;
    const float r2 = r * r;
    float s = fmaf(r2, 2.86567956e-6f, -1.98559923e-4f);
    s = fmaf(s, r2, 8.33338592e-3f);
    s = fmaf(s, r2, -1.66666672e-1f);
    s = fmaf(s * r2, r, r);
    float c = fmaf(r2, 2.44677067e-5f, -1.38877297e-3f);
    c = fmaf(c, r2, 4.16666567e-2f);
    c = fmaf(c, r2, -0.5f);
    c = fmaf(c, r2, 1.0f);
    float v = (n & 1) ? c : s;
    return (n & 2) ? -v : v;
}
#else
ADFP_DEV float adfp_sinf(float x) { return __builtin_amdgcn_sinf(adfp_turns(x)); }
#endif
ADFP_DEV void adfp_sincosf(float x, float& sn, float& cs) {
    const float t = adfp_turns(x);
    sn = __builtin_amdgcn_sinf(t);
    cs = __builtin_amdgcn_cosf(t);
}

// A sample position with a NaN coordinate (degenerate rays: 0/0 in the slab test, src/utils/Renderer.py:151) makes every
// Fourier feature NaN and the reference's decoders return NaN.  The integer relu below maps a NaN with the sign bit set
// to 0 (and v_max_f32 would drop any NaN), so the kernels restore the reference's answer at the output instead of paying a
// compare + select per hidden activation.
template <int NOUT>
ADFP_DEV void nan_point_outputs(const double pt[3], float* __restrict__ out) {
    const bool pnan = (pt[0] != pt[0]) | (pt[1] != pt[1]) | (pt[2] != pt[2]);
#pragma unroll
    for (int o = 0; o < NOUT; ++o) out[o] = pnan ? __builtin_nanf("") : out[o];
}

// relu as ONE integer instruction: max(bits(x), 0) keeps every positive float and maps every
// negative one (sign bit = negative int, -0.0 included) to +0.  fmaxf() on an MFMA result costs two
// VALU instructions (the compiler puts a canonicalising v_max in front of it, and folds fmed3 back
// into the same pair).  No inline asm on purpose: the compiler does not pad the MFMA-result hazard
// for an asm statement, and an asm v_max read half-finished accumulators.
ADFP_DEV float relu_f(float x) { const int b = __float_as_int(x); return __int_as_float(b > 0 ? b : 0); }

ADFP_DEV float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// inv_tsdf of mlp_tsdf.forward (decoder.py:244-248), all in f32 like the reference
ADFP_DEV float inv_tsdf(float t) {
    float s = 1.f - (t + 1.f) / 2.f;
    s = fminf(fmaxf(s, 0.f), 1.f);
    float u = -0.1f * logf((1.f / (s + 1e-8f)) - 1.f + 1e-7f);
    return fminf(fmaxf(u, -100.f), 100.f);
}

// ------------------------------------------------------------------------------------
// MFMA f32 32x32x2 chains.
//   D[row = out unit][col = point] += A[row][k] * B[k][col]
//   A operand: lane (i = l&31, h = l>>5) holds W[i][unit(s,h)]   (from the packed LDS image)
//   B operand: lane (p = l&31, h)        holds x[unit(s,h)] of point p
//   D: lane (p, h) reg r holds row kmapH(r,h) = (r&3) + 8*(r>>2) + 4*h of point p
// so a layer's 16 accumulator registers ARE the next layer's 16 B operands -- no lane
// movement, no LDS round trip.  ONE unit mapping is used everywhere (hidden units, Fourier
// features, grid channels): k-step s of lane-half h carries unit  32*(s>>4) + kmapH(s&15, h).
//
// Packed chain image of a [32 out x K in] block: K/4 step-groups; a step-group is 2 row-groups
// (h = 0,1) of 32 rows x 4 consecutive k-steps, each row-group padded from 128 to RG = 132
// floats.  Forward: lane reads one float4 per 4 k-steps (ds_read_b128, conflict-free).
// Backward (transposed operand W^T): lane j reads W[out][j] one float per MFMA
// (ds_read_b32); the padding makes the 32 lanes hit 32 different banks.
// ------------------------------------------------------------------------------------
#define ADFP_RG 132
#define ADFP_SG (2 * ADFP_RG)

__host__ __device__ constexpr int kmapH(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }
// unit carried by (k-step s, lane-half h)
__host__ __device__ constexpr int unit_of(int s, int h) { return 32 * (s >> 4) + kmapH(s & 15, h); }

template <int KS, typename BT>
ADFP_DEV void mfma_chain(f32x16& acc, const float* __restrict__ w, int lane_off, const BT& b, const int boff = 0) {
#pragma unroll
    for (int s4 = 0; s4 < KS / 4; ++s4) {
        const f32x4 a = *(const f32x4*)(w + s4 * ADFP_SG + lane_off);      // lane_off = h*RG + i*4
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[boff + 4 * s4 + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[boff + 4 * s4 + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[boff + 4 * s4 + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[boff + 4 * s4 + 3], acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting the next chain's LDS reads
}

// Transposed chain: acc[in unit j][point] += sum_out W[out][j] * g[out][point] for ONE 32-wide
// in-block of a [32 out x K in] image.  `w` = image base + the in-block's first step-group
// (16 k-steps = 4 step-groups per in-block); lane_off_t = ((j>>3)*2 + ((j>>2)&1))*RG + (j&3)
// for lane row j (the position of in-unit j inside its in-block); g = the 16 registers of the
// out-block's gradient in D layout.
template <typename BT>
ADFP_DEV void mfma_chain_T(f32x16& acc, const float* __restrict__ w, int lane_off_t, int h, const BT& g, const int goff = 0) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float a = w[lane_off_t + 4 * kmapH(s, 0) + 16 * h];              // row kmapH(s,h) = kmapH(s,0) + 4h
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, g[goff + s], acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// acc[r] = bias[row(r,h)]: rows 8q+4h .. 8q+4h+3 are one float4
ADFP_DEV void bias_init(f32x16& acc, const float* __restrict__ bias, int h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 t = *(const f32x4*)(bias + 8 * q + 4 * h);
        acc[4 * q + 0] = t.x; acc[4 * q + 1] = t.y; acc[4 * q + 2] = t.z; acc[4 * q + 3] = t.w;
    }
}
ADFP_DEV void relu_bias(f32x16& acc, const float* __restrict__ bias, int h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 t = *(const f32x4*)(bias + 8 * q + 4 * h);
        acc[4 * q + 0] = relu_f(acc[4 * q + 0]) + t.x;
        acc[4 * q + 1] = relu_f(acc[4 * q + 1]) + t.y;
        acc[4 * q + 2] = relu_f(acc[4 * q + 2]) + t.z;
        acc[4 * q + 3] = relu_f(acc[4 * q + 3]) + t.w;
    }
}
ADFP_DEV unsigned pos_mask(const f32x16& acc) {
    unsigned m = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) m |= (acc[r] > 0.f) ? (1u << r) : 0u;
    return m;
}

// ------------------------------------------------------------------------------------
// Layouts: flat (state_dict order) and packed (MFMA operand order) images of one decoder.
// MLP(dim=3, c_dim=CDIM, hidden=32, n_blocks=5, skips=[2]) decoder.py:110-166
// ------------------------------------------------------------------------------------
template <int CDIM, int NOUT>
struct DecLayout {
    static constexpr int EMB = 93;
    static constexpr int KSE = 48;            // k-steps covering the 93 (padded 96) Fourier features
    static constexpr int KSC = CDIM / 2;      // k-steps of fc_c
    __host__ __device__ static constexpr int in_dim(int i) { return i == 0 ? 93 : (i == 3 ? 125 : 32); }
    __host__ __device__ static constexpr int ks(int i) { return i == 0 ? KSE : (i == 3 ? KSE + 16 : 16); }
    __host__ __device__ static constexpr int chain_floats(int ksteps) { return (ksteps / 4) * ADFP_SG; }
    // ---- flat
    __host__ __device__ static constexpr int F_FC(int i) { return i * (32 * CDIM + 32); }
    static constexpr int F_EB = 5 * (32 * CDIM + 32);
    __host__ __device__ static constexpr int F_PL(int i) {
        int o = F_EB + 3 * EMB;
        for (int k = 0; k < i; ++k) o += 32 * in_dim(k) + 32;
        return o;
    }
    static constexpr int F_OW = F_PL(5);
    static constexpr int F_OB = F_OW + NOUT * 32;
    static constexpr int F_TOTAL = F_OB + NOUT;
    // ---- packed
    static constexpr int P_BM = 0;                                  // [96][4], feature-index order
    __host__ __device__ static constexpr int layer_floats(int i) { return chain_floats(ks(i)) + 32 + chain_floats(KSC) + 32; }
    __host__ __device__ static constexpr int P_WP(int i) {
        int o = 384;
        for (int k = 0; k < i; ++k) o += layer_floats(k);
        return o;
    }
    __host__ __device__ static constexpr int P_BP(int i) { return P_WP(i) + chain_floats(ks(i)); }
    __host__ __device__ static constexpr int P_WC(int i) { return P_BP(i) + 32; }
    __host__ __device__ static constexpr int P_BC(int i) { return P_WC(i) + chain_floats(KSC); }
    static constexpr int P_WO = P_WP(5);                             // [2][NOUT][16]
    static constexpr int P_BO = P_WO + 2 * NOUT * 16;                // [4]
    static constexpr int P_TOTAL = P_BO + 4;
};

// mlp_tsdf: 2 -> 64 -> 128 -> 128 -> 64 -> 2  (decoder.py:212-228)
struct AttLayout {
    // flat
    static constexpr int F_W0 = 0, F_B0 = 128;                        // [64][2], [64]
    static constexpr int F_W1 = 192, F_B1 = F_W1 + 128 * 64;          // [128][64]
    static constexpr int F_W2 = F_B1 + 128, F_B2 = F_W2 + 128 * 128;  // [128][128]
    static constexpr int F_W3 = F_B2 + 128, F_B3 = F_W3 + 64 * 128;   // [64][128]
    static constexpr int F_WO = F_B3 + 64, F_BO = F_WO + 2 * 64;      // [2][64]
    static constexpr int F_TOTAL = F_BO + 2;
    // packed
    static constexpr int BLK1 = (32 / 4) * ADFP_SG;                   // one 32-row out-block, K = 64
    static constexpr int BLK2 = (64 / 4) * ADFP_SG;                   // K = 128
    static constexpr int P_A0 = 0;                                    // [64][4] = (w0, w1, b, 0), unit order
    static constexpr int P_W1 = 256;                                  // 4 out-blocks
    static constexpr int P_B1 = P_W1 + 4 * BLK1;
    static constexpr int P_W2 = P_B1 + 128;                           // 4 out-blocks
    static constexpr int P_B2 = P_W2 + 4 * BLK2;
    static constexpr int P_W3 = P_B2 + 128;                           // 2 out-blocks
    static constexpr int P_B3 = P_W3 + 2 * BLK2;
    static constexpr int P_WO = P_B3 + 64;                            // [2 h][2 o][32]
    static constexpr int P_BO = P_WO + 128;
    static constexpr int P_TOTAL = P_BO + 4;
};


// ------------------------------------------------------------------------------------------
// staging rows (point-major, one row per point of the current chunk) for the weight gradients
// ------------------------------------------------------------------------------------------
template <int CDIM>
struct DecStage {
    static constexpr int SX = 0;                       // [x, y, z, 1, 0 ...]
    static constexpr int SE = 32;                      // Fourier features (96)
    static constexpr int SC = 128;                     // grid features (CDIM)
    __host__ __device__ static constexpr int SH(int i) { return 128 + CDIM + 32 * i; }     // h_0..h_4
    __host__ __device__ static constexpr int SGP(int i) { return SH(5) + 32 * i; }          // d/d pre_i
    __host__ __device__ static constexpr int SGH(int i) { return SGP(5) + 32 * i; }         // d/d h_i
    static constexpr int SGA = SGH(5);                 // d/d (p @ B) (96)
    static constexpr int SGO = SGA + 96;               // d/d out (32, first NOUT used)
    static constexpr int NCOLS = SGO + 32;
    // The f16-split backward keeps a row in two pieces: the decoder's INPUTS, columns [0, NX), are written by the training
    // forward (k_decode_h<..., TRAIN>) into caller-owned rows for all points; the GRADIENT blocks, columns [NX, NCOLS), by
    // k_decode_bwd_h into the chunked staging buffer.  k_outer_h puts the two pieces side by side in its LDS tile.
    static constexpr int NX = 128 + CDIM + 160;
    static constexpr int NG = NCOLS - NX;
    // What is actually stored of the two pieces (k_outer_h keeps the full column numbering in its LDS tile and fills in the
    // rest): the Fourier features are recomputed there from x, y, z (96 of the X piece's floats), and d/d pre_i = ReLU mask .
    // d/d h_i from the forward's mask words (160 of the G piece's).  Staging traffic is what bounds the weight-gradient path.
    static constexpr int NXM = 32 + CDIM + 160;       // [head: room for x, y, z, 1, 0 ..., NOT written by the training forward] | c | h_0..h_4
    static constexpr int NGM = 160 + 96 + 32;         // d/d h_0..h_4 | d/d (p @ B) | d/d out
    __host__ __device__ static constexpr int xm(int col) { return col < 32 ? col : col - 96; }     // X column -> offset in the stored piece
    __host__ __device__ static constexpr int gm(int col) { return col - SGH(0); }                  // G column (SGH.. on) -> offset
};
struct AttStage {
    static constexpr int AX = 0;                       // [occ_in, u, 1, 0 ...]
    static constexpr int AH0 = 32, AH1 = 96, AH2 = 224, AH3 = 352;
    static constexpr int AG0 = 416, AG1 = 480, AG2 = 608, AG3 = 736;
    static constexpr int AGL = 800;                    // d/d logits (2 used)
    static constexpr int NCOLS = 832;
};

// 16 registers of a D-layout block -> columns col + kmapH(r,h) of the point's staging row
template <typename VT>
ADFP_DEV void stage_block(float* __restrict__ row, int col, int h, const VT& v, const int voff = 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 t = {v[voff + 4 * q + 0], v[voff + 4 * q + 1], v[voff + 4 * q + 2], v[voff + 4 * q + 3]};
        *(f32x4*)(row + col + 8 * q + 4 * h) = t;
    }
}
// a block whose only non-zero entries are columns 0..3 (x, y, z, 1 / g_out / occ, u, 1)
ADFP_DEV void stage_head(float* __restrict__ row, int col, int h, f32x4 head) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 t = (q == 0 && h == 0) ? head : f32x4{0.f, 0.f, 0.f, 0.f};
        *(f32x4*)(row + col + 8 * q + 4 * h) = t;
    }
}


// ---------------------------------------------------------------------------------------------
// Wave-wide sums and an inclusive product scan on DPP (data-parallel primitives inside the VALU: no trip
// through the LDS crossbar that __shfl / ds_bpermute takes, ~8 cycles per step instead of ~100).
//   quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140,
//   row_shr:n = 0x110 + n, wave_shr:1 = 0x138 (gfx9 family)
// ---------------------------------------------------------------------------------------------
template <int CTRL, bool BOUND_CTRL = true>      // BOUND_CTRL false: lanes without a source keep `old`
ADFP_DEV float dpp_f32(float v, float old = 0.f) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, BOUND_CTRL));
}
template <int CTRL>
ADFP_DEV double dpp_f64(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
ADFP_DEV float readlane_f32(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
ADFP_DEV double readlane_f64(double v, int l) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// sum over the 64 lanes, valid in every lane
ADFP_DEV float wave_sum(float v) {
    v += dpp_f32<0xB1>(v); v += dpp_f32<0x4E>(v); v += dpp_f32<0x141>(v); v += dpp_f32<0x140>(v);      // every lane: its 16-lane row
    return (readlane_f32(v, 0) + readlane_f32(v, 16)) + (readlane_f32(v, 32) + readlane_f32(v, 48));
}
ADFP_DEV double wave_sum(double v) {
    v += dpp_f64<0xB1>(v); v += dpp_f64<0x4E>(v); v += dpp_f64<0x141>(v); v += dpp_f64<0x140>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
// inclusive product scan over the 64 lanes; `total` = product of all lanes
ADFP_DEV float wave_scan_mul(float v, int lane, float& total) {
    v *= dpp_f32<0x111, false>(v, 1.f); v *= dpp_f32<0x112, false>(v, 1.f);                            // inside each row of 16
    v *= dpp_f32<0x114, false>(v, 1.f); v *= dpp_f32<0x118, false>(v, 1.f);
    const float t0 = readlane_f32(v, 15), t1 = readlane_f32(v, 31), t2 = readlane_f32(v, 47), t3 = readlane_f32(v, 63);
    const float p1 = t0, p2 = t0 * t1, p3 = p2 * t2;
    const int row = lane >> 4;
    v *= row == 0 ? 1.f : (row == 1 ? p1 : (row == 2 ? p2 : p3));
    total = p3 * t3;
    return v;
}

// ---------------------------------------------------------------------------------------------
// Tile hand-out with a CHIP-WIDE tail (round 4).  Per-wave end stamps of the fused decoder launch (tools/phase_g.py, spans-only
// build) showed the waves of a workgroup ending within 23 us of each other but the WORKGROUPS 1 352 ... 1 444 us after the start
// (one XCD 3.4 % behind the others): with every workgroup owning the same number of tiles the launch waits for its slowest CU
// while the others idle -- 4.9 % of the launch.  So a workgroup owns only its first `j_static` slots (whole rows of the fixed
// split); the tiles behind them (from `pool_base` on) are handed out in CHUNKS of NWW consecutive tiles from ONE device counter,
// one global atomic per chunk -- issued by the wave that DRAWS the chunk's first slot, a tile ahead of its use, and published to
// the workgroup through a small LDS ring -- so a fast CU takes more chunks.  (Round 2 handed the last eighth out tile by tile:
// 25 000 returning atomics on one address made the launch 18 % slower; this is 1 in 12 of that.)
// `pool` == NULL: the fixed split (bench hooks without a workspace).  *pool must be zero at launch.
// ---------------------------------------------------------------------------------------------
// ring size: an entry is overwritten ADFP_POOL_RING chunks later; a wave uses the slot it drew one tile earlier, and in one tile of the
// slowest wave (the arbiter starves the youngest wave of a SIMD to ~0.45 x the oldest one's rate) the workgroup draws < 3 chunks
#define ADFP_POOL_RING 64
struct TilePlan { int j_static, pool_base; };
__host__ __device__ inline TilePlan tile_plan(int ntiles, int nwg, int nww, bool pooled) {
    const int per_row = nwg * nww, rows = (ntiles + per_row - 1) / per_row;
    int keep = rows;
    if (pooled && rows >= 6) { int tail = rows / 10; if (tail < 2) tail = 2; keep = rows - tail; }
    TilePlan p; p.j_static = keep * nww; p.pool_base = keep * per_row;
    if (!pooled || rows < 6) { p.j_static = 0x7fffffff; p.pool_base = 0; }
    return p;
}
// The wait for a ring entry is BOUNDED: the entry is published by a sibling wave a tile ahead of its use, so a poll normally matches at
// once; should it never match (a counter block that was not zero at launch, an entry overwritten early), the wave raises
// ADFP_STATUS_POOL_TIMEOUT in `status` after ~2^22 polls and leaves its tile loop -- an error the host sees, not a hung GPU.
template <int NWW>
ADFP_DEV int claim_tile_pool(int& j, int* s_next, unsigned long long* s_ring, const TilePlan plan, int ntiles, int* pool, int* status = nullptr) {
    int tile;
    if (j < plan.j_static) tile = blockIdx.x * NWW + (j % NWW) + (j / NWW) * (gridDim.x * NWW);
    else {
        const int d = j - plan.j_static, c = d / NWW, slot = d - c * NWW;
        unsigned long long e;
        int polls = 0;
        for (;;) {
            e = __hip_atomic_load(s_ring + (c & (ADFP_POOL_RING - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((int)(e >> 32) == c + 1) break;
            if (++polls > (1 << 22)) {
                if (status && (threadIdx.x & 63) == 0) __hip_atomic_fetch_or(status, 32 /* ADFP_STATUS_POOL_TIMEOUT */, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return -1;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        tile = plan.pool_base + (int)(unsigned)e + slot;
    }
    if (tile >= ntiles) return -1;
    int jn = 0;
    if ((threadIdx.x & 63) == 0) {
        jn = atomicAdd(s_next, 1);
        if (jn >= plan.j_static) {
            const int d = jn - plan.j_static, c = d / NWW;
            if (d - c * NWW == 0) {                      // this draw opens chunk c: fetch its tiles now, a tile before anyone needs them
                const int base = atomicAdd(pool, NWW);
                __hip_atomic_store(s_ring + (c & (ADFP_POOL_RING - 1)), ((unsigned long long)(unsigned)(c + 1) << 32) | (unsigned)base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    j = __builtin_amdgcn_readfirstlane(jn);
    return tile;
}

// ---------------------------------------------------------------------------------------------
// Tile hand-out inside a workgroup.  A workgroup of NWW waves owns the tiles
//   wg_tile(j) = blockIdx.x * NWW + (j % NWW) + (j / NWW) * (gridDim.x * NWW),   j = 0, 1, 2, ...
// (the same set a fixed wave-strided split would give it) and its waves take slots j from an LDS
// ticket.  The SIMD arbiter favours the oldest wave, so with a fixed split the young waves of every
// SIMD finished ~20 % after the old ones (measured per-wave end stamps, tools/ab_stage.py).
// *s_next must be initialised to NWW before the barrier that precedes the loop; wave w starts at j = w.
// Returns the tile of slot j (or -1 when the workgroup's tiles are exhausted) and advances j.
// ---------------------------------------------------------------------------------------------
template <int NWW>
ADFP_DEV int claim_tile(int& j, int* s_next, int ntiles) {
#ifdef ADFP_XCD_TILES
    // XCD-aware variant: workgroup b runs on XCD b % 8 (round-robin dispatch); give every XCD one CONTIGUOUS
    // eighth of the tiles (a band of image rows) so that its private L2 sees one eighth of the frustum's grid
    // lines instead of a 1-in-8 sample of all of them.
    int tile;
    if ((gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7, bl = blockIdx.x >> 3, gl = gridDim.x >> 3;
        const int per = ((ntiles + 7) / 8 + NWW - 1) / NWW * NWW;            // tiles per XCD, whole slots
        const int loc = bl * NWW + (j % NWW) + (j / NWW) * (gl * NWW);
        if (loc >= per) return -1;
        tile = xcd * per + loc;
    } else tile = blockIdx.x * NWW + (j % NWW) + (j / NWW) * (gridDim.x * NWW);
#else
    const int tile = blockIdx.x * NWW + (j % NWW) + (j / NWW) * (gridDim.x * NWW);
#endif
    if (tile >= ntiles) return -1;
#ifdef ADFP_STATIC_TILES        // A/B switch: the fixed split
    j += NWW;
#else
    int jn = 0;
    if ((threadIdx.x & 63) == 0) jn = atomicAdd(s_next, 1);
    j = __builtin_amdgcn_readfirstlane(jn);
#endif
    return tile;
}

// A packed weight image (n4 pieces of 16 bytes) into LDS by the NT threads of the workgroup, B loads per thread in flight.  The plain
// copy loop -- lds[i] = src[i], i += NT -- compiles to one load, s_waitcnt vmcnt(0), one ds_write per trip: a memory latency per
// 16 bytes and thread.  The attention network's training forward (132 KB image, 256 threads: 33 trips) spent 14.3 us of its 44
// there, before its first tile (tools/experiments/att_span.py); every kernel that keeps an image in LDS paid ~0.4 us per trip.
// Call before the __syncthreads() that publishes the image.  (The exact-f32 kernels of adfp_kernels.hip / adfp_backward.h keep the plain
// loop: their workgroups are persistent over thousands of tiles in the one place they are timed, value_exact_f32_mode.)
template <int NT, int B = 16>
ADFP_DEV void image_to_lds(void* __restrict__ lds, const void* __restrict__ src, int n4) {
    typedef unsigned piece __attribute__((ext_vector_type(4)));
    const piece* __restrict__ s4 = (const piece*)src;
    piece* __restrict__ d4 = (piece*)lds;
    for (int base = (int)threadIdx.x; base < n4; base += B * NT) {
        piece t[B];
#pragma unroll
        for (int b = 0; b < B; ++b) { const int i = base + b * NT; t[b] = s4[i < n4 ? i : n4 - 1]; }      // unconditional: no branch between the loads
#pragma unroll
        for (int b = 0; b < B; ++b) { const int i = base + b * NT; if (i < n4) d4[i] = t[b]; }
    }
}
// the same for an image whose size is known at compile time: as few equal batches as the registers at the head of a kernel allow
// (up to 34 pieces = 136 registers per thread in a 256-thread workgroup, which runs one wave per SIMD; 17 otherwise)
template <int NT, int N4>
ADFP_DEV void image_to_lds(void* __restrict__ lds, const void* __restrict__ src) {
    constexpr int TRIPS = (N4 + NT - 1) / NT, MAXB = NT <= 256 ? 34 : 17, NBATCH = (TRIPS + MAXB - 1) / MAXB, B = (TRIPS + NBATCH - 1) / NBATCH;
    image_to_lds<NT, B>(lds, src, N4);
}

// out[0] = max(parts[0 .. n)) by ONE workgroup of NT threads (all of them call)
template <int NT>
ADFP_DEV void max_fold_block(const float* __restrict__ parts, int n, float* __restrict__ out) {
    __shared__ float s_m[NT / 64];
    float mx = 0.f;
    for (int i = threadIdx.x; i < n; i += NT) mx = fmaxf(mx, parts[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = s_m[0];
        for (int w = 1; w < NT / 64; ++w) m = fmaxf(m, s_m[w]);
        *out = m;
    }
}
