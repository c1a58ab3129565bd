// Does the f32 -> f16 conversion of the operand split (v_cvt_pkrtz_f16_f32, v_fma_mixlo_f16) leave a trace in the wave's sticky
// IEEE exception status (TRAPSTS.EXCP) when a value is outside the f16 range?  If it does, the range guard of the f16-split
// decoders is ONE s_getreg per wave instead of a v_max3 per pair of split values.
//   hipcc --offload-arch=gfx950 -O2 -o trapsts_overflow trapsts_overflow.hip && ./trapsts_overflow
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out, unsigned* trap) {
    const int t = threadIdx.x;
    unsigned before, after_cvt, after_mix;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 32)" : "=s"(before));
    const float a = in[blockIdx.x * 64 + t];
    const h2 hp = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a, a));
    unsigned u = __builtin_bit_cast(unsigned, hp);
    asm volatile("s_nop 8\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 32)" : "=s"(after_cvt) : "v"(u));
    unsigned m1 = 0xBC00BC00u;
    asm volatile("" : "+v"(m1));
    const h2 neg1 = __builtin_bit_cast(h2, m1);
    h2 lp;
    lp[0] = (_Float16)__builtin_fmaf((float)hp[0], (float)neg1[0], a);
    lp[1] = (_Float16)__builtin_fmaf((float)hp[1], (float)neg1[0], a * 3.0f);
    unsigned v = __builtin_bit_cast(unsigned, lp);
    asm volatile("s_nop 8\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 32)" : "=s"(after_mix) : "v"(v));
    out[(blockIdx.x * 64 + t) * 2] = u; out[(blockIdx.x * 64 + t) * 2 + 1] = v;
    if (t == 0) { trap[blockIdx.x * 3] = before; trap[blockIdx.x * 3 + 1] = after_cvt; trap[blockIdx.x * 3 + 2] = after_mix; }
}
int main() {
    const float vals[6] = {1.0f, 60000.0f, 65504.0f, 65520.0f, 1.0e5f, 3.0e5f};
    float h_in[6 * 64]; unsigned h_out[6 * 64 * 2], h_trap[18];
    for (int b = 0; b < 6; ++b) for (int t = 0; t < 64; ++t) h_in[b * 64 + t] = (t == 5) ? vals[b] : 0.5f;     // ONE lane holds the value
    float* d_in; unsigned *d_out, *d_trap;
    hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, sizeof(h_out)); hipMalloc(&d_trap, sizeof(h_trap));
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(6), dim3(64), 0, 0, d_in, d_out, d_trap);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost); hipMemcpy(h_trap, d_trap, sizeof(h_trap), hipMemcpyDeviceToHost);
    for (int b = 0; b < 6; ++b)
        printf("x = %9.1f  hi|hi = %08x  lo = %08x   TRAPSTS before %08x  after cvt_pkrtz %08x  after fma_mixlo/hi %08x\n", vals[b],
               h_out[(b * 64 + 5) * 2], h_out[(b * 64 + 5) * 2 + 1], h_trap[b * 3], h_trap[b * 3 + 1], h_trap[b * 3 + 2]);
    return 0;
}
