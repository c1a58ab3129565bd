#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#pragma clang diagnostic ignored "-Wunused-value"
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, float* hi, float* lo, float* hi2, float* lo2, int n) {
    int i = (blockIdx.x * blockDim.x + threadIdx.x) * 2; if (i >= n) return;
    const float a = x[i], b = x[i + 1];
    const f16x2 hp = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
    hi[i] = (float)hp[0]; hi[i + 1] = (float)hp[1];
    lo[i] = (float)(_Float16)(a - (float)hp[0]); lo[i + 1] = (float)(_Float16)(b - (float)hp[1]);
    const float ah = __uint_as_float(__float_as_uint(a) & 0xFFFFE000u), bh = __uint_as_float(__float_as_uint(b) & 0xFFFFE000u);
    const f16x2 h2 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(ah, bh));
    const f16x2 l2 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a - ah, b - bh));
    hi2[i] = (float)h2[0]; hi2[i + 1] = (float)h2[1]; lo2[i] = (float)l2[0]; lo2[i + 1] = (float)l2[1];
}
int main() {
    const int n = 1 << 20; float *x, *h, *l, *h2, *l2; hipMalloc(&x, n * 4); hipMalloc(&h, n * 4); hipMalloc(&l, n * 4); hipMalloc(&h2, n * 4); hipMalloc(&l2, n * 4);
    float* hx = (float*)malloc(n * 4); float* a = (float*)malloc(n * 4); float* b = (float*)malloc(n * 4); float* c = (float*)malloc(n * 4); float* d = (float*)malloc(n * 4);
    const float scales[6] = {1.f, 100.f, 3e4f, 0.01f, 1e-4f, 1e-6f};
    for (int t = 0; t < 6; ++t) {
        srand(t); for (int i = 0; i < n; ++i) hx[i] = ((float)rand() / RAND_MAX * 2 - 1) * scales[t];
        hipMemcpy(x, hx, n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(n / 512), dim3(256), 0, 0, x, h, l, h2, l2, n);
        hipMemcpy(a, h, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b, l, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(c, h2, n * 4, hipMemcpyDeviceToHost); hipMemcpy(d, l2, n * 4, hipMemcpyDeviceToHost);
        double e1 = 0, e2 = 0; int bad = 0;
        for (int i = 0; i < n; ++i) { double r1 = fabs((double)hx[i] - ((double)a[i] + b[i])), r2 = fabs((double)hx[i] - ((double)c[i] + d[i]));
            if (!std::isfinite(a[i]) || !std::isfinite(b[i])) bad++;
            e1 = fmax(e1, r1 / scales[t]); e2 = fmax(e2, r2 / scales[t]); }
        printf("scale %g: new split max |x-(hi+lo)|/scale %.3e (nonfinite %d)   old split %.3e\n", scales[t], e1, bad, e2);
    }
    return 0;
}
