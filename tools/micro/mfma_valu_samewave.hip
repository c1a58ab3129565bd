// Microbenchmark: inside ONE wave, how many independent VALU instructions issue in the shadow of a
// dependent MFMA chain?  Loop body = 1 MFMA (dependent on the previous one) + NV v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, int KIND>   // KIND 0: f32 32x32x2, 1: bf16 32x32x16, 2: f32 16x16x4
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v[16]; for (int j = 0; j < 16; ++j) v[j] = a + j;
    bf16x8 x, y; for (int r = 0; r < 8; ++r) { x[r] = 0x3f80; y[r] = 0x3f80; }
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            else if (KIND == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
            else acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j & 15] = fmaf(v[j & 15], b, a);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    float s = 0; for (int j = 0; j < 16; ++j) s += v[j];
    for (int r = 0; r < 16; ++r) s += acc[r];
    s += acc4.x;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 3 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int NV, int KIND> void run(float* out, long long* cyc, const char* name) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<NV, KIND>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s NV=%2d: %.1f cycles per (MFMA + NV VALU)\n", name, NV, (double)h / (iters * 8));
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    run<0, 0>(out, cyc, "f32 32x32x2 "); run<4, 0>(out, cyc, "f32 32x32x2 "); run<8, 0>(out, cyc, "f32 32x32x2 ");
    run<12, 0>(out, cyc, "f32 32x32x2 "); run<16, 0>(out, cyc, "f32 32x32x2 "); run<24, 0>(out, cyc, "f32 32x32x2 ");
    run<0, 1>(out, cyc, "bf16 32x32x16"); run<4, 1>(out, cyc, "bf16 32x32x16"); run<8, 1>(out, cyc, "bf16 32x32x16"); run<16, 1>(out, cyc, "bf16 32x32x16");
    run<0, 2>(out, cyc, "f32 16x16x4 "); run<4, 2>(out, cyc, "f32 16x16x4 "); run<8, 2>(out, cyc, "f32 16x16x4 "); run<16, 2>(out, cyc, "f32 16x16x4 ");
    return 0;
}
