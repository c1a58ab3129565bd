// Microbenchmark (gfx950): does the SIMD cost of v_fma_f32 / v_fmac_f32 depend on which VGPR banks (index mod 4) its operands
// sit in?  Explicit registers: 16 independent accumulator chains v16..v31, multiplicand / addend registers chosen per variant.
// Prints cycles per instruction per SIMD at W = 1 / 2 / 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CLOB "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31"
#define X16(OP, A, B) \
    OP(16, A, B) OP(17, A, B) OP(18, A, B) OP(19, A, B) OP(20, A, B) OP(21, A, B) OP(22, A, B) OP(23, A, B) \
    OP(24, A, B) OP(25, A, B) OP(26, A, B) OP(27, A, B) OP(28, A, B) OP(29, A, B) OP(30, A, B) OP(31, A, B)
#define FMA3(D, A, B) "v_fma_f32 v" #D ", v" #D ", v" #A ", v" #B "\n"
#define FMAC(D, A, B) "v_fmac_f32 v" #D ", v" #A ", v" #B "\n"
#define FMAS(D, A, B) "v_fma_f32 v" #D ", v" #D ", s" #A ", v" #B "\n"
#define FMACS(D, A, B) "v_fmac_f32 v" #D ", s" #A ", v" #B "\n"
#define FMAK(D, A, B) "v_fma_f32 v" #D ", v" #D ", 0.5, v" #B "\n"
#define MUL(D, A, B) "v_mul_f32 v" #D ", v" #D ", v" #A "\n"
#define MAXI(D, A, B) "v_max_i32 v" #D ", v" #D ", v" #A "\n"
#define MAXF(D, A, B) "v_max_f32 v" #D ", v" #D ", v" #A "\n"
#define SUBF(D, A, B) "v_sub_f32 v" #D ", v" #D ", v" #A "\n"
#define CVTPK(D, A, B) "v_cvt_pkrtz_f16_f32 v" #D ", v" #D ", v" #A "\n"
#define RND(D, A, B) "v_rndne_f32 v" #D ", v" #D "\n"
#define MOV(D, A, B) "v_mov_b32 v" #D ", v" #A "\n"
#define ANDB(D, A, B) "v_and_b32 v" #D ", v" #D ", v" #A "\n"
#define ADDU(D, A, B) "v_add_u32 v" #D ", v" #D ", v" #A "\n"
#define LSHL(D, A, B) "v_lshlrev_b32 v" #D ", 1, v" #D "\n"
#define FMAMIX(D, A, B) "v_fma_mix_f32 v" #D ", v" #A ", v" #B ", v" #D " op_sel_hi:[1,1,0]\n"
#define MAX3(D, A, B) "v_max3_f32 v" #D ", v" #D ", v" #A ", v" #B "\n"

#define KERNEL(NAME, OP, A, B)                                                                                      \
    __global__ __launch_bounds__(1024) void NAME(float* out, int iters, long long* stamps) {                          \
        asm volatile("v_mov_b32 v2, 1.0\nv_mov_b32 v3, 0.5\nv_mov_b32 v4, 1.0\nv_mov_b32 v5, 0.5\nv_mov_b32 v6, 1.0\n"        \
                     "v_mov_b32 v7, 0.5\nv_mov_b32 v8, 1.0\nv_mov_b32 v9, 0.5\ns_mov_b32 s20, 1.0\n" ::: CLOB, "s20");     \
        asm volatile(X16(MOV, 2, 2) ::: CLOB);                                                                            \
        __syncthreads();                                                                                                  \
        const long long t0 = __builtin_amdgcn_s_memtime();                                                                \
        for (int i = 0; i < iters; ++i) asm volatile(X16(OP, A, B) X16(OP, A, B) X16(OP, A, B) X16(OP, A, B) ::: CLOB);   \
        const long long t1 = __builtin_amdgcn_s_memtime();                                                                \
        float s;                                                                                                          \
        asm volatile("v_add_f32 %0, v16, v31" : "=v"(s)::CLOB);                                                           \
        out[blockIdx.x * 1024 + threadIdx.x] = s;                                                                         \
        if (blockIdx.x == 7 && (threadIdx.x & 63) == 0) stamps[threadIdx.x >> 6] = t1 - t0;                               \
    }

KERNEL(k_fma3_b23, FMA3, 2, 3)      // multiplicand bank 2, addend bank 3, accumulators cycle through banks 0..3
KERNEL(k_fma3_b48, FMA3, 4, 8)      // both in bank 0
KERNEL(k_fma3_b22, FMA3, 2, 2)      // the same register twice
KERNEL(k_fmac_b23, FMAC, 2, 3)
KERNEL(k_fmac_b48, FMAC, 4, 8)
KERNEL(k_fmas, FMAS, 20, 3)         // SGPR multiplicand
KERNEL(k_fmacs, FMACS, 20, 3)
KERNEL(k_fmak, FMAK, 2, 3)          // inline constant
KERNEL(k_mul, MUL, 2, 3)
KERNEL(k_maxi, MAXI, 2, 3)
KERNEL(k_maxf, MAXF, 2, 3)
KERNEL(k_subf, SUBF, 2, 3)
KERNEL(k_cvtpk, CVTPK, 2, 3)
KERNEL(k_rnd, RND, 2, 3)
KERNEL(k_and, ANDB, 2, 3)
KERNEL(k_addu, ADDU, 2, 3)
KERNEL(k_lshl, LSHL, 2, 3)
KERNEL(k_fmamix, FMAMIX, 2, 3)
KERNEL(k_max3, MAX3, 2, 3)

static long long* g_stamps;
static float* g_out;
template <typename K>
void run(const char* name, K kern) {
    double r[3];
    for (int W = 1; W <= 3; ++W) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipMemset(g_stamps, 0, 16 * 8);
            hipLaunchKernelGGL(kern, dim3(256), dim3(256 * W), 0, 0, g_out, 3000, g_stamps);
            (void)hipDeviceSynchronize();
        }
        long long h[16]; (void)hipMemcpy(h, g_stamps, 16 * 8, hipMemcpyDeviceToHost);
        double mx = 0; for (int w = 0; w < 4 * W; ++w) mx = h[w] > mx ? h[w] : mx;
        r[W - 1] = mx / (64.0 * 3000) / W;
    }
    printf("%-44s per SIMD %6.2f %6.2f %6.2f\n", name, r[0], r[1], r[2]);
}
int main() {
    (void)hipMalloc(&g_out, 256 * 1024 * 4); (void)hipMalloc(&g_stamps, 16 * 8);
    printf("SIMD cycles per instruction at W = 1 / 2 / 3 waves per SIMD (16 independent chains v16..v31)\n");
    run("v_fma_f32 d,d,v2,v3 (banks 2,3)", k_fma3_b23); run("v_fma_f32 d,d,v4,v8 (banks 0,0)", k_fma3_b48);
    run("v_fma_f32 d,d,v2,v2", k_fma3_b22); run("v_fmac_f32 d,v2,v3", k_fmac_b23); run("v_fmac_f32 d,v4,v8", k_fmac_b48);
    run("v_fma_f32 d,d,s20,v3", k_fmas); run("v_fmac_f32 d,s20,v3", k_fmacs); run("v_fma_f32 d,d,0.5,v3", k_fmak);
    run("v_mul_f32 d,d,v2", k_mul); run("v_max_i32 d,d,v2", k_maxi); run("v_max_f32 d,d,v2", k_maxf); run("v_sub_f32 d,d,v2", k_subf);
    run("v_cvt_pkrtz_f16_f32 d,d,v2", k_cvtpk); run("v_rndne_f32 d,d", k_rnd); run("v_and_b32 d,d,v2", k_and);
    run("v_add_u32 d,d,v2", k_addu); run("v_lshlrev_b32 d,1,d", k_lshl); run("v_fma_mix_f32 d,v2,v3,d", k_fmamix);
    run("v_max3_f32 d,d,v2,v3", k_max3);
    return 0;
}
