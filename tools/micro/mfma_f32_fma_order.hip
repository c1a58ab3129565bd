// Does v_mfma_f32_32x32x2_f32 form its K = 2 sum as two f32 fma steps in k order?  Role P of k_decode_bwd_roles computes
// p @ embedder._B (K = 3) with two such instructions and relies on the result being the forward's
//     fmaf(z, bz, fmaf(y, by, x * bx))
// bit for bit (the Fourier features of the backward are then the forward's).  Counts the elements of a 32 x 32 product over many
// random draws whose bits differ from that expression, with arguments of the decoder's size (|p| <= 1, B ~ N(0, 25)).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o mfma_f32_fma_order mfma_f32_fma_order.hip && ./mfma_f32_fma_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const float* __restrict__ pts, const float* __restrict__ B, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
    const float* P = pts + 96 * blockIdx.x;
    const float* Bm = B + 96 * blockIdx.x;
    const float x = P[3 * p], y = P[3 * p + 1], z = P[3 * p + 2];
    const float bx = Bm[3 * p], by = Bm[3 * p + 1], bz = Bm[3 * p + 2];
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? y : x, h ? by : bx, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? 0.f : z, h ? 0.f : bz, acc, 0, 0, 0);
    // D: lane (j = p, h) register r = row (r & 3) + 8 (r >> 2) + 4 h = point, column = feature j
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        out[1024 * blockIdx.x + 32 * row + p] = acc[r];
    }
}

int main() {
    const int nb = 4096;
    std::vector<float> pts(96 * nb), B(96 * nb), out(1024 * nb);
    srand(7);
    auto uni = [] { return (float)rand() / (float)RAND_MAX; };
    for (auto& v : pts) v = 2.f * uni() - 1.f;
    for (auto& v : B) { const float u1 = uni() + 1e-7f, u2 = uni(); v = 25.f * sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }
    float *dp, *db, *dout;
    hipMalloc(&dp, pts.size() * 4); hipMalloc(&db, B.size() * 4); hipMalloc(&dout, out.size() * 4);
    hipMemcpy(dp, pts.data(), pts.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, dp, db, dout);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    long long diff = 0, total = 0; double worst = 0.0;
    for (int b = 0; b < nb; ++b)
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                const float* P = &pts[96 * b + 3 * i];
                const float* Bm = &B[96 * b + 3 * j];
                const float ref = fmaf(P[2], Bm[2], fmaf(P[1], Bm[1], P[0] * Bm[0]));
                const float got = out[1024 * b + 32 * i + j];
                ++total;
                if (memcmp(&ref, &got, 4) != 0) { ++diff; const double e = fabs((double)ref - got); if (e > worst) worst = e; }
            }
    printf("v_mfma_f32_32x32x2_f32, K = 3 as (x, y) then (z, 0), against fmaf(z, bz, fmaf(y, by, x * bx)): %lld of %lld elements differ in their bits; largest difference %.3e\n",
           diff, total, worst);
    return 0;
}
