// Microbenchmark: accuracy of three range-reduced sine variants for |x| up to 3000 rad
//  A: pi/2 reduction + sin/cos minimax pair (the kernels' adfp_sinf)
//  B: pi reduction + one odd degree-11 polynomial + sign flip
//  C: 2*pi reduction by fma + hardware v_sin_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#pragma clang diagnostic ignored "-Wunused-value"
__device__ float sinA(float x) {
    const float k = rintf(x * 0.636619772f);
    float r = fmaf(k, -1.57079601e+00f, x); r = fmaf(k, -3.13916473e-07f, r); r = fmaf(k, -5.39030253e-15f, r);
    const int n = (int)k; const float r2 = r * r;
    float s = fmaf(r2, 2.86567956e-6f, -1.98559923e-4f); s = fmaf(s, r2, 8.33338592e-3f); s = fmaf(s, r2, -1.66666672e-1f); s = fmaf(s * r2, r, r);
    float c = fmaf(r2, 2.44677067e-5f, -1.38877297e-3f); c = fmaf(c, r2, 4.16666567e-2f); c = fmaf(c, r2, -0.5f); c = fmaf(c, r2, 1.0f);
    float v = (n & 1) ? c : s; return (n & 2) ? -v : v;
}
__device__ float sinB(float x) {
    const float k = rintf(x * 0.318309886f);
    float r = fmaf(k, -3.14159203e+00f, x); r = fmaf(k, -6.27832947e-07f, r); r = fmaf(k, -1.07806051e-14f, r);
    const float r2 = r * r;
    float q = fmaf(r2, -2.3846689956030787e-08f, 2.752261934801936e-06f);
    q = fmaf(q, r2, -0.00019840804452542216f); q = fmaf(q, r2, 0.008333330042660236f); q = fmaf(q, r2, -0.1666666716337204f);
    const float s = fmaf(r * r2, q, r);
    return __uint_as_float(__float_as_uint(s) ^ ((unsigned)(int)k << 31));
}
__device__ float sinC(float x) {
    const float k = rintf(x * 0.159154943f);
    float r = fmaf(k, -6.28318405e+00f, x); r = fmaf(k, -1.25566589e-06f, r); r = fmaf(k, -2.15612101e-14f, r);
    return __builtin_amdgcn_sinf(r * 0.159154943f);
}
__device__ float sinD(float x) {
    const float k = rintf(x * 0.15915494f);
    float t = fmaf(x, 0.15915494f, -k);
    t = fmaf(x, 6.4206382432985265e-09f, t);
    return __builtin_amdgcn_sinf(t);
}
__global__ void k(const float* x, float* a, float* b, float* c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    a[i] = sinA(x[i]); b[i] = sinD(x[i]); c[i] = sinC(x[i]);
}
int main() {
    const int n = 1 << 22; float *x, *a, *b, *c; hipMalloc(&x, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4);
    float* hx = (float*)malloc(n * 4); float* h = (float*)malloc(n * 4);
    srand(3); for (int i = 0; i < n; ++i) hx[i] = ((float)rand() / RAND_MAX * 2 - 1) * 3000.f;
    hipMemcpy(x, hx, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, a, b, c, n);
    float* outs[3] = {a, b, c}; const char* names[3] = {"A pi/2 + sin/cos polys", "D turns by 2 fma + v_sin", "C 2pi + v_sin_f32      "};
    for (int m = 0; m < 3; ++m) {
        hipMemcpy(h, outs[m], n * 4, hipMemcpyDeviceToHost);
        double e = 0; for (int i = 0; i < n; ++i) e = fmax(e, fabs((double)h[i] - sin((double)hx[i])));
        printf("%s max abs err %.3e\n", names[m], e);
    }
    return 0;
}
