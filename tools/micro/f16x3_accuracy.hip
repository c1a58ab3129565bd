// Microbenchmark: accuracy of a 3-product f16 split (hi*hi + hi*lo + lo*hi) on
// v_mfma_f32_32x32x16_f16 against exact f32 MFMA and a double reference, including small
// operands (are f16 subnormal MFMA inputs flushed?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// A [32][K], B [K][32] row-major f32; K multiple of 16.  One wave.
template <int MODE>   // 0: f32 MFMA, 1: f16x3 unscaled, 2: f16x3 with lo scaled by 2^11, 3: plain f16 (hi only)
__global__ void k(const float* A, const float* B, float* D, int K) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc, accx;
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accx[r] = 0.f; }
    if (MODE == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + h], B[(k + h) * 32 + i], acc, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 16) {
            f16x8 ah, al, bh, bl;
            for (int j = 0; j < 8; ++j) {
                const float a = A[i * K + k0 + 8 * h + j], b = B[(k0 + 8 * h + j) * 32 + i];
                const float ahf = __uint_as_float(__float_as_uint(a) & 0xFFFFE000u), bhf = __uint_as_float(__float_as_uint(b) & 0xFFFFE000u);
                ah[j] = (_Float16)ahf; bh[j] = (_Float16)bhf;
                const float sc = MODE == 2 ? 2048.f : 1.f;
                al[j] = (_Float16)((a - ahf) * sc); bl[j] = (_Float16)((b - bhf) * sc);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            if (MODE == 1) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
            } else if (MODE == 2) {
                accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accx, 0, 0, 0);
                accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accx, 0, 0, 0);
            }
        }
        if (MODE == 2) for (int r = 0; r < 16; ++r) acc[r] = fmaf(accx[r], 1.f / 2048.f, acc[r]);
    }
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

int main() {
    const int K = 128;
    float *A, *B, *D; hipMalloc(&A, 32 * K * 4); hipMalloc(&B, K * 32 * 4); hipMalloc(&D, 32 * 32 * 4);
    float hA[32 * K], hB[K * 32], hD[1024];
    const float scalesA[4] = {1.f, 0.03f, 1e-3f, 5.f}, scalesB[4] = {0.3f, 0.2f, 0.3f, 0.3f};
    for (int t = 0; t < 4; ++t) {
        srand(7 + t);
        for (int n = 0; n < 32 * K; ++n) { hA[n] = ((float)rand() / RAND_MAX * 2 - 1) * scalesA[t]; hB[n] = ((float)rand() / RAND_MAX * 2 - 1) * scalesB[t]; }
        hipMemcpy(A, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(B, hB, sizeof(hB), hipMemcpyHostToDevice);
        double ref[1024], scale = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)hA[i * K + k] * hB[k * 32 + j]; ref[i * 32 + j] = s; scale = fmax(scale, fabs(s)); }
        const char* names[4] = {"f32 mfma      ", "f16x3 unscaled", "f16x3 scaled  ", "f16 hi only   "};
        for (int m = 0; m < 4; ++m) {
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, A, B, D, K);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, A, B, D, K);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, A, B, D, K);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, A, B, D, K);
            hipMemcpy(hD, D, sizeof(hD), hipMemcpyDeviceToHost);
            double e = 0; for (int n = 0; n < 1024; ++n) e = fmax(e, fabs(hD[n] - ref[n]));
            printf("|A|~%g |B|~%g  %s max abs err %.3e  (rel to max |ref| %.3e)\n", scalesA[t], scalesB[t], names[m], e, e / scale);
        }
    }
    return 0;
}
