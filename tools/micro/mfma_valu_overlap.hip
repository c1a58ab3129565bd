// Microbenchmark: do f32-input MFMAs (v_mfma_f32_32x32x2_f32) overlap with f32 VALU work of a
// co-resident wave on the same SIMD, or do they share the datapath?
// 512-thread workgroups, one per CU: waves 0-3 and 4-7 pair up on the 4 SIMDs.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int PRIO = 0>   // 0: all MFMA f32, 1: all VALU, 2: waves 0-3 MFMA f32 + waves 4-7 VALU, 3: all MFMA bf16, 4: bf16 + VALU
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* stamps = nullptr) {
    const long long t0 = clock64();
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = MODE == 0 || MODE == 3 || ((MODE == 2 || MODE == 4 || MODE == 5) && wave < 4);
    if (MODE == 5 && wave >= 4) return;
    if (MODE == 6 && wave < 4) return;
    const bool bf = MODE == 3 || MODE == 4;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
    if (do_mfma) {
        if (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
        if (!bf) {
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        } else {
            bf16x8 x, y; for (int r = 0; r < 8; ++r) { x[r] = 0x3f80; y[r] = 0x3f80; }
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int j = 0; j < 32; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
            }
        }
    } else {
        if (PRIO < 0) __builtin_amdgcn_s_setprio(-PRIO);
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {       // 256 independent-ish fmas per iteration
                v0 = fmaf(v0, b, a); v1 = fmaf(v1, b, a); v2 = fmaf(v2, b, a); v3 = fmaf(v3, b, a);
                v4 = fmaf(v4, b, a); v5 = fmaf(v5, b, a); v6 = fmaf(v6, b, a); v7 = fmaf(v7, b, a);
            }
        }
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (stamps && blockIdx.x == 7 && (threadIdx.x & 63) == 0) stamps[threadIdx.x >> 6] = clock64() - t0;
}

static long long* g_stamps = nullptr;
template <int MODE, int PRIO = 0>
float run(float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, PRIO>), dim3(256), dim3(512), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, PRIO>), dim3(256), dim3(512), 0, 0, out, iters, g_stamps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; hipMemcpy(h, g_stamps, 64, hipMemcpyDeviceToHost);
    printf("   per-wave Mcycles:"); for (int i = 0; i < 8; ++i) printf(" %.2f", h[i] * 1e-6); printf("\n");
    return ms;
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&g_stamps, 64); hipMemset(g_stamps, 0, 64);
    const int iters = 2000;
    // MFMA f32: 16 per iter * 64 cyc = 1024 cyc/iter/wave; 8 waves -> 2 per SIMD -> 2048 cyc per iter per SIMD
    // VALU: 256 fma per iter per wave
    printf("all MFMA f32      : %.3f ms\n", run<0>(out, iters));
    printf("all VALU          : %.3f ms\n", run<1>(out, iters));
    printf("half MFMA f32 + half VALU : %.3f ms  (separate pipes -> ~max(half,half); shared -> ~sum)\n", run<2>(out, iters));
    printf("half MFMA f32 (setprio 3) + half VALU : %.3f ms\n", run<2, 3>(out, iters));
    printf("half MFMA f32 + half VALU (setprio 3 on VALU waves): %.3f ms\n", run<2, -3>(out, iters));
    printf("MFMA f32 waves only (others exit): %.3f ms\n", run<5>(out, iters));
    printf("VALU waves only (others exit)    : %.3f ms\n", run<6>(out, iters));
    printf("all MFMA bf16     : %.3f ms\n", run<3>(out, iters));
    printf("half MFMA bf16 + half VALU: %.3f ms\n", run<4>(out, iters));
    printf("half MFMA bf16 (setprio 3) + half VALU: %.3f ms\n", run<4, 3>(out, iters));
    return 0;
}
