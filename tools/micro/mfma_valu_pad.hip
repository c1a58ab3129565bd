// Microbenchmark: can a partner wave's VALU work issue inside the gaps of an f16 MFMA stream
// (v_mfma_f32_32x32x16_f16) on the SAME SIMD when the MFMA wave pads its stream with s_nop
// (i.e. does not present a stalled MFMA to the issue port), and how many of the MFMA wave's OWN
// VALU instructions hide per MFMA gap?
// 512-thread workgroups, one per CU: waves 0-3 (MFMA) and 4-7 (VALU) pair up on the 4 SIMDs.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// PAD: number of "s_nop 3" (4 cycles each) after every MFMA.  FILL: own v_fma per MFMA gap.
// PARTNER: 1 = waves 4-7 run the VALU loop, 0 = they exit.
template <int PAD, int FILL, int PARTNER>
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* stamps) {
    const long long t0 = clock64();
    const int wave = threadIdx.x >> 6;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
    if (wave < 4) {
        f16x8 x, y; for (int r = 0; r < 8; ++r) { x[r] = (_Float16)1.0f; y[r] = (_Float16)1.0f; }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < PAD; ++p) asm volatile("s_nop 3");
                if (FILL > 0) v0 = fmaf(v0, b, a);
                if (FILL > 1) v1 = fmaf(v1, b, a);
                if (FILL > 2) v2 = fmaf(v2, b, a);
                if (FILL > 3) v3 = fmaf(v3, b, a);
                if (FILL > 4) v4 = fmaf(v4, b, a);
                if (FILL > 5) v5 = fmaf(v5, b, a);
                if (FILL > 6) v6 = fmaf(v6, b, a);
                if (FILL > 7) v7 = fmaf(v7, b, a);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        if (!PARTNER) return;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {       // 128 independent-ish fmas per iteration (4 per partner MFMA)
                v0 = fmaf(v0, b, a); v1 = fmaf(v1, b, a); v2 = fmaf(v2, b, a); v3 = fmaf(v3, b, a);
                v4 = fmaf(v4, b, a); v5 = fmaf(v5, b, a); v6 = fmaf(v6, b, a); v7 = fmaf(v7, b, a);
            }
        }
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((blockIdx.x == 7) && (threadIdx.x & 63) == 0) stamps[threadIdx.x >> 6] = clock64() - t0;
}

static long long* g_stamps = nullptr;
template <int PAD, int FILL, int PARTNER>
void run(float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<PAD, FILL, PARTNER>), dim3(256), dim3(512), 0, 0, out, iters, g_stamps);
    hipDeviceSynchronize();
    hipMemset(g_stamps, 0, 64);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<PAD, FILL, PARTNER>), dim3(256), dim3(512), 0, 0, out, iters, g_stamps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; hipMemcpy(h, g_stamps, 64, hipMemcpyDeviceToHost);
    const double nm = 32.0 * iters;
    printf("pad %d x4cyc, own fill %d, partner %d: %.3f ms | MFMA wave %.1f cyc/MFMA | VALU wave %.2f cyc/fma (%.1f fma per partner MFMA slot)\n",
           PAD, FILL, PARTNER, ms, h[0] / nm, PARTNER ? h[4] / (128.0 * iters) : 0.0,
           PARTNER ? (h[0] / nm) / (h[4] / (128.0 * iters)) : 0.0);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&g_stamps, 64);
    const int iters = 2000;
    run<0, 0, 0>(out, iters);
    run<0, 0, 1>(out, iters);
    run<2, 0, 1>(out, iters);
    run<4, 0, 1>(out, iters);
    run<6, 0, 1>(out, iters);
    run<8, 0, 1>(out, iters);
    run<0, 2, 0>(out, iters);
    run<0, 4, 0>(out, iters);
    run<0, 5, 0>(out, iters);
    run<0, 6, 0>(out, iters);
    run<0, 8, 0>(out, iters);
    run<0, 4, 1>(out, iters);
    return 0;
}
