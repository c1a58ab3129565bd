// Microbenchmark (gfx950): what does a VALU instruction cost next to v_mfma_f32_32x32x16_f16 streams, as a function of
//   F  = independent VALU fillers placed in every MFMA gap of the SAME wave (0..14; inline asm, so no SLP packing),
//   W  = waves per SIMD that run this same mixed stream (1, 2, 3; workgroup = 256 * W threads, one per CU),
//   PK = fillers are v_pk_fma_f32 (two f32 fmas per instruction) instead of v_fma_f32,
//   DEP = the MFMAs of a wave form ONE dependent accumulator chain (as in the decoders) or rotate over 4 accumulators.
// Prints SIMD cycles per MFMA-gap of ONE wave (s_memtime over the loop / MFMAs) and the derived cost per filler.
// The decoders' question: with ~14 VALU per MFMA, is time  MFMA + VALU  (serial) or  max(MFMA, VALU issue)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int F, int PK, int DEP>
__global__ __launch_bounds__(1024) void k(float* out, int iters, long long* stamps) {
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v[16];
    f32x2 pv[16];
    for (int j = 0; j < 16; ++j) { v[j] = a + j; pv[j] = f32x2{a + j, a - j}; }
    const f32x2 pb = {b, b}, pa = {a, a};
    f16x8 x, y;
    for (int r = 0; r < 8; ++r) { x[r] = (_Float16)(1.0f + 0.001f * (threadIdx.x & 7)); y[r] = (_Float16)0.5f; }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int q = DEP ? 0 : (j & 3);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < F; ++f) {                        // asm: the compiler would SLP-pack scalar fmas into v_pk_fma_f32
                if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pv[f]) : "v"(pb), "v"(pa));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[f]) : "v"(b), "v"(a));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += v[j] + pv[j][0] + pv[j][1];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (blockIdx.x == 7 && (threadIdx.x & 63) == 0) stamps[threadIdx.x >> 6] = t1 - t0;
}

// VALU only: N independent chains of v_fma_f32 / v_pk_fma_f32, W waves per SIMD
template <int PK>
__global__ __launch_bounds__(1024) void kv(float* out, int iters, long long* stamps) {
    const float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v[16]; f32x2 pv[16];
    for (int j = 0; j < 16; ++j) { v[j] = a + j; pv[j] = f32x2{a + j, a - j}; }
    const f32x2 pb = {b, b}, pa = {a, a};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pv[j]) : "v"(pb), "v"(pa));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(b), "v"(a));
            }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += v[j] + pv[j][0] + pv[j][1];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (blockIdx.x == 7 && (threadIdx.x & 63) == 0) stamps[threadIdx.x >> 6] = t1 - t0;
}

static long long* g_stamps = nullptr;
static float* g_out = nullptr;

template <int F, int PK, int DEP>
double run(int W, int iters) {
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(g_stamps, 0, 16 * 8);
        hipLaunchKernelGGL((k<F, PK, DEP>), dim3(256), dim3(256 * W), 0, 0, g_out, iters, g_stamps);
        hipDeviceSynchronize();
    }
    long long h[16]; hipMemcpy(h, g_stamps, 16 * 8, hipMemcpyDeviceToHost);
    double mx = 0; for (int w = 0; w < 4 * W; ++w) mx = h[w] > mx ? h[w] : mx;
    return mx / (16.0 * iters);          // s_memtime ticks (100 MHz? shader clock on gfx950: see header line) per MFMA gap of one wave
}
template <int PK>
double runv(int W, int iters) {
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(g_stamps, 0, 16 * 8);
        hipLaunchKernelGGL((kv<PK>), dim3(256), dim3(256 * W), 0, 0, g_out, iters, g_stamps);
        hipDeviceSynchronize();
    }
    long long h[16]; hipMemcpy(h, g_stamps, 16 * 8, hipMemcpyDeviceToHost);
    double mx = 0; for (int w = 0; w < 4 * W; ++w) mx = h[w] > mx ? h[w] : mx;
    return mx / (64.0 * iters);
}

template <int PK, int DEP>
void sweep(int iters) {
    printf("---- fillers = %s, MFMA chain %s\n", PK ? "v_pk_fma_f32" : "v_fma_f32", DEP ? "dependent (1 accumulator)" : "4 accumulators");
    printf("  F : cycles per MFMA gap of one wave at W = 1 / 2 / 3 waves per SIMD  (SIMD cycles per gap = value / W)\n");
#define ROW(F) printf(" %2d : %7.1f %7.1f %7.1f   | per SIMD %6.1f %6.1f %6.1f\n", F, run<F, PK, DEP>(1, iters), run<F, PK, DEP>(2, iters), run<F, PK, DEP>(3, iters), \
                      run<F, PK, DEP>(1, iters), run<F, PK, DEP>(2, iters) / 2, run<F, PK, DEP>(3, iters) / 3);
    ROW(0) ROW(2) ROW(4) ROW(5) ROW(6) ROW(8) ROW(10) ROW(12) ROW(14)
#undef ROW
}

int main() {
    hipMalloc(&g_out, 256 * 1024 * 4); hipMalloc(&g_stamps, 16 * 8);
    const int iters = 4000;
    printf("VALU only, cycles per instruction of one wave at W = 1/2/3/4 waves per SIMD:\n");
    printf("  v_fma_f32    : %5.2f %5.2f %5.2f %5.2f\n", runv<0>(1, iters), runv<0>(2, iters), runv<0>(3, iters), runv<0>(4, iters));
    printf("  v_pk_fma_f32 : %5.2f %5.2f %5.2f %5.2f\n", runv<1>(1, iters), runv<1>(2, iters), runv<1>(3, iters), runv<1>(4, iters));
    sweep<0, 1>(iters);
    sweep<0, 0>(iters);
    sweep<1, 1>(iters);
    return 0;
}
