// Microbenchmark (gfx950): SIMD issue cost of the VALU instructions the f16-split decoder is made of, one instruction kind per
// kernel, 16 independent chains, W = 1 / 2 / 3 waves per SIMD.  Prints cycles per instruction of ONE wave (s_memtime) and the
// per-SIMD cost (that / W).  Question: the decoder's VALU mix averages ~4.2 cycles of SIMD time per instruction where a plain
// v_fma_f32 costs 2.5 -- which instructions are the expensive ones?
#include <hip/hip_runtime.h>
#include <cstdio>

#define BODY(NAME, ASM, TYPE, INIT)                                                                            \
    __global__ __launch_bounds__(1024) void NAME(float* out, int iters, long long* stamps) {                   \
        float a = threadIdx.x * 1e-3f + 1.f, b = 1.0001f;                                                       \
        double da = a, db = b;                                                                                  \
        unsigned ua = threadIdx.x + 3, ub = 77;                                                                 \
        (void)da; (void)db; (void)ua; (void)ub;                                                                 \
        TYPE v[16];                                                                                             \
        for (int j = 0; j < 16; ++j) v[j] = INIT;                                                               \
        __syncthreads();                                                                                        \
        const long long t0 = __builtin_amdgcn_s_memtime();                                                      \
        for (int i = 0; i < iters; ++i) {                                                                       \
            _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                       \
                _Pragma("unroll") for (int j = 0; j < 16; ++j) { ASM; }                                         \
        }                                                                                                       \
        const long long t1 = __builtin_amdgcn_s_memtime();                                                      \
        float s = 0.f;                                                                                          \
        for (int j = 0; j < 16; ++j) s += (float)v[j];                                                          \
        out[blockIdx.x * 1024 + threadIdx.x] = s;                                                               \
        if (blockIdx.x == 7 && (threadIdx.x & 63) == 0) stamps[threadIdx.x >> 6] = t1 - t0;                     \
    }

BODY(k_fma, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(b), "v"(a)), float, a + j)
BODY(k_fmac, asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[j]) : "v"(b), "v"(a)), float, a + j)
BODY(k_mul, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j]) : "v"(b)), float, a + j)
BODY(k_add, asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(b)), float, a + j)
BODY(k_maxi, asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[j]) : "v"(ub)), unsigned, ua + j)
BODY(k_max3, asm volatile("v_max3_f32 %0, |%0|, |%1|, %2" : "+v"(v[j]) : "v"(b), "v"(a)), float, a + j)
BODY(k_rndne, asm volatile("v_rndne_f32 %0, %0" : "+v"(v[j])), float, a + j)
BODY(k_sin, asm volatile("v_sin_f32 %0, %0" : "+v"(v[j])), float, 0.01f * (a + j))
BODY(k_cvtpk, asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[j]) : "v"(b)), float, a + j)
BODY(k_fmamix, asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(v[j]) : "v"(ua), "v"(ub)), float, a + j)
BODY(k_fma64, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[j]) : "v"(db), "v"(da)), double, da + j)
BODY(k_mul64, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[j]) : "v"(db)), double, da + j)
BODY(k_add64, asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[j]) : "v"(db)), double, da + j)
BODY(k_mullo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[j]) : "v"(ub)), unsigned, ua + j)
BODY(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(ub)), unsigned, ua + j)
BODY(k_lshladd64, asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(v[j]) : "v"((unsigned long long)ub)), unsigned long long, (unsigned long long)(ua + j))
BODY(k_pkfma, asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(db), "v"(da)), double, da + j)

static long long* g_stamps;
static float* g_out;
template <typename K>
void run(const char* name, K kern) {
    double r[3];
    for (int W = 1; W <= 3; ++W) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(g_stamps, 0, 16 * 8);
            hipLaunchKernelGGL(kern, dim3(256), dim3(256 * W), 0, 0, g_out, 3000, g_stamps);
            hipDeviceSynchronize();
        }
        long long h[16]; hipMemcpy(h, g_stamps, 16 * 8, hipMemcpyDeviceToHost);
        double mx = 0; for (int w = 0; w < 4 * W; ++w) mx = h[w] > mx ? h[w] : mx;
        r[W - 1] = mx / (64.0 * 3000);
    }
    printf("%-22s per wave %6.2f %6.2f %6.2f   per SIMD %6.2f %6.2f %6.2f\n", name, r[0], r[1], r[2], r[0], r[1] / 2, r[2] / 3);
}
int main() {
    hipMalloc(&g_out, 256 * 1024 * 4); hipMalloc(&g_stamps, 16 * 8);
    printf("cycles per instruction at W = 1 / 2 / 3 waves per SIMD\n");
    run("v_fma_f32", k_fma); run("v_fmac_f32", k_fmac); run("v_mul_f32", k_mul); run("v_add_f32", k_add);
    run("v_max_i32", k_maxi); run("v_max3_f32 |a| |b|", k_max3); run("v_rndne_f32", k_rndne); run("v_sin_f32", k_sin);
    run("v_cvt_pkrtz_f16_f32", k_cvtpk); run("v_fma_mix_f32", k_fmamix);
    run("v_fma_f64", k_fma64); run("v_mul_f64", k_mul64); run("v_add_f64", k_add64);
    run("v_mul_lo_u32", k_mullo); run("v_cndmask_b32", k_cndmask); run("v_lshl_add_u64", k_lshladd64); run("v_pk_fma_f32", k_pkfma);
    return 0;
}
