// Microbenchmark (gfx950): the decoder's per-tile loop on the two f16 MFMA shapes, same output tile per wave (32 units x 32 points
// per layer), same 3-product operand split, same LDS weight image size and reads, same VALU work per tile (gather and
// Fourier-feature stand-ins with the decoder's instruction mix, relu + range guard + split per layer), random data, 3 waves per
// SIMD (768-thread workgroup, one per CU, 130 KB of LDS) -- the launch shape of k_decode_lc.
//
//   SHAPE 0: v_mfma_f32_32x32x16_f16   90 MFMAs per network-tile (30 k-steps x 3 products), 32 pipe cycles each
//   SHAPE 1: v_mfma_f32_16x16x32_f16  180 MFMAs per network-tile (15 K=32 groups x 2 out-blocks x 2 point-blocks x 3 products), 16 each
//
// Question (MI355X_MICROARCH.md "DVFS give-back" item 7, cdna_hip_programming.md rule 28): bare MFMA loops on random data hold a
// higher clock on the 16x16x32 form (1.12-1.15 x FLOP/s at equal cycles).  Does a loop with the decoder's ~14 VALU instructions per
// 32x32x16 MFMA gain from it?  Each MFMA also holds the SIMD's vector issue for ~8 cycles whatever its shape, and the 16x16x32 form
// needs twice as many.  Reported: wall time per launch (HIP events, A-B-A), wave cycles per tile (s_memtime) and the in-kernel clock
// (s_memtime / s_memrealtime).
//
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -o mfma_shape_ab mfma_shape_ab.hip && ./mfma_shape_ab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define DEV __device__ __forceinline__

constexpr int IMG_WORDS = 16640;           // ~65 KB: one decoder's split image (30 k-steps x 512 words + Fourier / bias rows)
constexpr int NT = 768;

DEV float relu_f(float v) { return __int_as_float(max(__float_as_int(v), 0)); }

// the product library's split8 (adfp_decode_h.h): 8 f32 -> 8 f16 hi + 8 f16 lo, running |max| for the range guard
DEV void split8(const float* __restrict__ x, f16x8& hi, f16x8& lo, float& amax) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    u32x4 uh, ul;
    unsigned m1 = 0xBC00BC00u;
    asm volatile("" : "+v"(m1));
    const h2 neg1 = __builtin_bit_cast(h2, m1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = x[2 * j], b = x[2 * j + 1];
        amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), amax);
        const h2 hp = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a, b));
        h2 lp;
        lp[0] = (_Float16)__builtin_fmaf((float)hp[0], (float)neg1[0], a);
        lp[1] = (_Float16)__builtin_fmaf((float)hp[1], (float)neg1[0], b);
        uh[j] = __builtin_bit_cast(unsigned, hp);
        ul[j] = __builtin_bit_cast(unsigned, lp);
    }
    hi = __builtin_bit_cast(f16x8, uh);
    lo = __builtin_bit_cast(f16x8, ul);
    asm volatile("" : "+v"(amax));
}

DEV float sin_turns(float x) {               // the decoders' sine: exact reduction in turns, then v_sin_f32
    const float t = __builtin_fmaf(x, 0.15915494f, 0.f);
    const float r = (t + 12582912.f) - 12582912.f;
    return __builtin_amdgcn_sinf(__builtin_fmaf(x, 0.15915494f, -r));
}

// ---- the per-point front end both variants share: a gather stand-in (8 corners x 16 channels of fmac from an L2-resident table)
// and 48 Fourier features (3 fma + reduction + v_sin each), split into MFMA operands
DEV void front_end(const float* __restrict__ table, int lane, float px, float py, float pz, const float* __restrict__ brow,
                   f16x8 ch[2], f16x8 cl[2], f16x8 eh[6], f16x8 el[6], float& amax) {
    float c[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) c[j] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float w = __builtin_fmaf(px, 0.25f * k, py) * pz;
        const f32x4* g = (const f32x4*)(table + ((lane * 8 + k) & 1023) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = g[q];
            c[4 * q] = __builtin_fmaf(w, v.x, c[4 * q]); c[4 * q + 1] = __builtin_fmaf(w, v.y, c[4 * q + 1]);
            c[4 * q + 2] = __builtin_fmaf(w, v.z, c[4 * q + 2]); c[4 * q + 3] = __builtin_fmaf(w, v.w, c[4 * q + 3]);
        }
    }
    split8(c, ch[0], cl[0], amax);
    split8(c + 8, ch[1], cl[1], amax);
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 bm = *(const f32x4*)(brow + (8 * ks + j) * 4);
            e[j] = sin_turns(__builtin_fmaf(pz, bm.z, __builtin_fmaf(py, bm.y, px * bm.x)));
        }
        float dummy = 0.f;
        split8(e, eh[ks], el[ks], dummy);
    }
}

// ---- SHAPE 0: 32x32x16.  A k-step = 512 words: [hi|lo][h][32 rows][8 halves]; lane (p, h) reads its hi and lo with two ds_read_b128
template <int NK>
DEV void chain32(f32x16& acc, const unsigned* __restrict__ w, const f16x8* __restrict__ xh, const f16x8* __restrict__ xl) {
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const f16x8 ah = __builtin_bit_cast(f16x8, *(const u32x4*)(w + ks * 512));
        const f16x8 al = __builtin_bit_cast(f16x8, *(const u32x4*)(w + ks * 512 + 256));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh[ks], acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

DEV void tile32(const unsigned* __restrict__ img, const float* __restrict__ table, int lane, float px, float py, float pz, float& amax, float* out) {
    const int p = lane & 31, h = lane >> 5;
    const unsigned* wl = img + h * 128 + p * 4;
    f16x8 ch[2], cl[2], eh[6], el[6];
    front_end(table, lane, px, py, pz, (const float*)img + 15360 + 16 * h, ch, cl, eh, el, amax);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc;
    f16x8 hh[2], hl[2];
    int off = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const f32x4* b = (const f32x4*)((const float*)img + 15872 + 32 * i + 4 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const f32x4 t = b[2 * q]; acc[4 * q] = t.x; acc[4 * q + 1] = t.y; acc[4 * q + 2] = t.z; acc[4 * q + 3] = t.w; }
        if (i == 0) { chain32<6>(acc, wl + off * 512, eh, el); off += 6; }
        else if (i == 3) { chain32<6>(acc, wl + off * 512, eh, el); off += 6; chain32<2>(acc, wl + off * 512, hh, hl); off += 2; }
        else { chain32<2>(acc, wl + off * 512, hh, hl); off += 2; }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = relu_f(acc[r]) + 0.01f;
        chain32<2>(acc, wl + off * 512, ch, cl); off += 2;
        if (i < 4) {
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = acc[r] * 0.05f;          // keeps the random network's activations O(1)
            split8(t, hh[0], hl[0], amax);
            split8(t + 8, hh[1], hl[1], amax);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s = __builtin_fmaf(acc[r], 0.01f * (r + 1), s);
    s += __shfl_xor(s, 32);
    *out = s;
}

// ---- SHAPE 1: 16x16x32.  A K=32 group = 1024 words: [out-block 0|1][hi|lo][4 k-groups][16 rows][8 halves]; lane (n, g) reads 4 x
// ds_read_b128 per group (the same bytes as two 32x32x16 k-steps); B operands per point-block: the lane's 8 values of that block
template <int NG>
DEV void chain16(f32x4 acc[2][2], const unsigned* __restrict__ w, const f16x8 (*xh)[2], const f16x8 (*xl)[2]) {
#pragma unroll
    for (int kg = 0; kg < NG; ++kg) {
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            const f16x8 ah = __builtin_bit_cast(f16x8, *(const u32x4*)(w + kg * 1024 + ob * 512));
            const f16x8 al = __builtin_bit_cast(f16x8, *(const u32x4*)(w + kg * 1024 + ob * 512 + 256));
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                acc[ob][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh[kg][pb], acc[ob][pb], 0, 0, 0);
                acc[ob][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl[kg][pb], acc[ob][pb], 0, 0, 0);
                acc[ob][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh[kg][pb], acc[ob][pb], 0, 0, 0);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

DEV void tile16(const unsigned* __restrict__ img, const float* __restrict__ table, int lane, float px, float py, float pz, float& amax, float* out) {
    const int n = lane & 15, g = lane >> 4;
    const unsigned* wl = img + g * 64 + n * 4;
    // the lane serves TWO points (one per point-block) with 8 units each: the front end's work per lane is the same 16 gathered
    // channels + 48 Fourier features, arranged as 2 points x (8 channels + 24 features); the stand-in computes the same instruction
    // mix once and deals the operands out to the two blocks
    f16x8 ch[2], cl[2], eh[6], el[6];
    front_end(table, lane, px, py, pz, (const float*)img + 15360 + 16 * (g & 1), ch, cl, eh, el, amax);
    __builtin_amdgcn_sched_barrier(0);
    f16x8 eh2[3][2], el2[3][2], ch2[1][2], cl2[1][2], hh[1][2], hl[1][2];
#pragma unroll
    for (int k = 0; k < 3; ++k) { eh2[k][0] = eh[2 * k]; eh2[k][1] = eh[2 * k + 1]; el2[k][0] = el[2 * k]; el2[k][1] = el[2 * k + 1]; }
    ch2[0][0] = ch[0]; ch2[0][1] = ch[1]; cl2[0][0] = cl[0]; cl2[0][1] = cl[1];
    f32x4 acc[2][2];
    int off = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const f32x4* b = (const f32x4*)((const float*)img + 15872 + 32 * i + 4 * g);
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) { acc[ob][0] = b[4 * ob]; acc[ob][1] = b[4 * ob]; }
        if (i == 0) { chain16<3>(acc, wl + off * 1024, eh2, el2); off += 3; }
        else if (i == 3) { chain16<3>(acc, wl + off * 1024, eh2, el2); off += 3; chain16<1>(acc, wl + off * 1024, hh, hl); off += 1; }
        else { chain16<1>(acc, wl + off * 1024, hh, hl); off += 1; }
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ob][pb][r] = relu_f(acc[ob][pb][r]) + 0.01f;
        chain16<1>(acc, wl + off * 1024, ch2, cl2); off += 1;
        if (i < 4) {
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                float t[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { t[r] = acc[0][pb][r] * 0.05f; t[4 + r] = acc[1][pb][r] * 0.05f; }
                split8(t, hh[0][pb], hl[0][pb], amax);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int r = 0; r < 4; ++r) s = __builtin_fmaf(acc[ob][pb][r], 0.01f * (4 * ob + r + 1), s);
    s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
    *out = s;
}

template <int SHAPE>
__global__ __launch_bounds__(NT, NT / 256) void k(const unsigned* __restrict__ image, const float* __restrict__ table, float* __restrict__ out,
                                                  int tiles_per_wave, long long* __restrict__ stamps) {
    __shared__ __attribute__((aligned(16))) unsigned lds[2 * IMG_WORDS];         // two images like k_decode_lc: 130 KB, one workgroup per CU
    for (int i = threadIdx.x; i < 2 * IMG_WORDS / 4; i += NT) ((u32x4*)lds)[i] = ((const u32x4*)image)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float amax = 0.f, acc_out = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < tiles_per_wave; ++t) {
        const float px = 0.37f * lane + 0.01f * t, py = 1.3f - 0.002f * lane, pz = 0.5f + 0.003f * (wave + t);
#pragma unroll
        for (int net = 0; net < 2; ++net) {                 // low, then colour: two images
            int o = net * IMG_WORDS;
            asm volatile("" : "+v"(o));
            float r;
            if (SHAPE == 0) tile32(lds + o, table, lane, px + net, py, pz, amax, &r);
            else tile16(lds + o, table, lane, px + net, py, pz, amax, &r);
            acc_out += r;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[(long long)blockIdx.x * NT + threadIdx.x] = acc_out + (amax > 60000.f ? 1.f : 0.f);
    if (lane == 0) { stamps[2 * (blockIdx.x * (NT / 64) + wave)] = t1 - t0; stamps[2 * (blockIdx.x * (NT / 64) + wave) + 1] = r1 - r0; }
}

int main() {
    int dev = 0; hipDeviceProp_t prop; hipGetDeviceProperties(&prop, dev);
    const int ncu = prop.multiProcessorCount;
    std::vector<unsigned> himg(2 * IMG_WORDS);
    srand(1);
    for (auto& w : himg) {                                  // random f16 pairs in [-0.5, 0.5): the sign / exponent / mantissa bits toggle
        auto h = [](float x) { _Float16 v = (_Float16)x; unsigned short u; __builtin_memcpy(&u, &v, 2); return (unsigned)u; };
        w = h((rand() / (float)RAND_MAX - 0.5f)) | (h((rand() / (float)RAND_MAX - 0.5f)) << 16);
    }
    for (int net = 0; net < 2; ++net)                        // the f32 rows (Fourier matrix, biases) of each image
        for (int i = 15360; i < IMG_WORDS; ++i) { float v = (rand() / (float)RAND_MAX - 0.5f) * (i < 15872 ? 50.f : 0.2f); __builtin_memcpy(&himg[net * IMG_WORDS + i], &v, 4); }
    std::vector<float> htab(1024 * 16);
    for (auto& v : htab) v = rand() / (float)RAND_MAX - 0.5f;
    unsigned* image; float* table; float* out; long long* stamps;
    hipMalloc(&image, himg.size() * 4); hipMalloc(&table, htab.size() * 4); hipMalloc(&out, (size_t)ncu * NT * 4); hipMalloc(&stamps, (size_t)ncu * (NT / 64) * 16);
    hipMemcpy(image, himg.data(), himg.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(table, htab.data(), htab.size() * 4, hipMemcpyHostToDevice);
    const int tiles = 600;                                   // per wave: the frame's 614 400 tiles over 256 CUs x 12 waves = 200; x3 for a longer launch
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int shape, int reps) {
        float best = 1e30f, sum = 0.f;
        for (int r = 0; r < reps; ++r) {
            hipEventRecord(e0);
            if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(ncu), dim3(NT), 0, 0, image, table, out, tiles, stamps);
            else hipLaunchKernelGGL(k<1>, dim3(ncu), dim3(NT), 0, 0, image, table, out, tiles, stamps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, ms); sum += ms;
        }
        std::vector<long long> hs((size_t)ncu * (NT / 64) * 2);
        hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> cyc, clk;
        for (size_t i = 0; i < hs.size() / 2; ++i) { cyc.push_back((double)hs[2 * i] / tiles); clk.push_back((double)hs[2 * i] / hs[2 * i + 1] * 0.1); }
        std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
        printf("  %-24s wall %.3f ms avg / %.3f best of %d   wave cycles per tile (both networks) median %.0f   in-kernel clock median %.2f GHz\n",
               shape == 0 ? "32x32x16 (180 MFMA/tile)" : "16x16x32 (360 MFMA/tile)", sum / reps, best, reps, cyc[cyc.size() / 2], clk[clk.size() / 2]);
        return sum / reps;
    };
    printf("decoder-shaped loop, %d CUs x %d threads (3 waves / SIMD), %d tiles per wave, random data\n", ncu, NT, tiles);
    // ~2 s of back-to-back launches first (the clock settles under load), then A - B - A - B
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(k<0>, dim3(ncu), dim3(NT), 0, 0, image, table, out, tiles, stamps);
    hipDeviceSynchronize();
    double a1 = run(0, 20), b1 = run(1, 20), a2 = run(0, 20), b2 = run(1, 20);
    printf("16x16x32 / 32x32x16 wall: %.3f (first pair), %.3f (second pair)\n", b1 / a1, b2 / a2);
    return 0;
}
