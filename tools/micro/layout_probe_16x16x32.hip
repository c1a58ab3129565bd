// Probe (gfx950): operand / result register layouts of v_mfma_f32_16x16x32_f16 and the lane mapping of v_permlane32_swap_b32,
// printed as facts the 16x16x32 decoder kernel (csrc/adfp_decode_g.h) is built on.
//   hipcc -O3 --offload-arch=gfx950 -o layout_probe_16x16x32 layout_probe_16x16x32.hip && ./layout_probe_16x16x32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// A[i][k] = (i == ai && k == ak), B[k][n] = (k == bk && n == bn) under the ASSUMED layouts
//   A: lane l holds row l % 16, k = 8 (l / 16) + j;   B: lane l holds column l % 16, k = 8 (l / 16) + j
// -> D must be 1 exactly at (ai, bn) when ak == bk; report where it lands: D assumed lane l = column l % 16, rows 4 (l / 16) + r
__global__ void probe(int ai, int ak, int bk, int bn, float* out) {
    const int l = threadIdx.x;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)((l % 16 == ai && 8 * (l / 16) + j == ak) ? 1.f : 0.f);
        b[j] = (_Float16)((l % 16 == bn && 8 * (l / 16) + j == bk) ? 1.f : 0.f);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
}
__global__ void swap_probe(unsigned* out) {
    const unsigned l = threadIdx.x;
    unsigned x = 100 + l, y = 200 + l;
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    out[2 * l] = r[0]; out[2 * l + 1] = r[1];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4);
    float h[256];
    int bad = 0;
    for (int t = 0; t < 64; ++t) {
        const int ai = (t * 7 + 3) % 16, bn = (t * 5 + 1) % 16, k = (t * 11 + 2) % 32;
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, ai, k, k, bn, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            const float want = (l % 16 == bn && 4 * (l / 16) + r == ai) ? 1.f : 0.f;
            if (h[l * 4 + r] != want) { if (bad < 5) printf("MISMATCH t=%d lane %d reg %d: %g (want %g)\n", t, l, r, h[l * 4 + r], want); ++bad; }
        }
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, ai, k, (k + 1) % 32, bn, d);      // different k: no product
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int i = 0; i < 256; ++i) if (h[i] != 0.f) { if (bad < 5) printf("MISMATCH (k differs) t=%d idx %d: %g\n", t, i, h[i]); ++bad; }
    }
    printf("v_mfma_f32_16x16x32_f16 layouts (A: row l%%16, k 8(l/16)+j; B: col l%%16, k 8(l/16)+j; D: col l%%16, rows 4(l/16)+r): %s\n", bad ? "WRONG" : "confirmed");
    unsigned* u; hipMalloc(&u, 128 * 4);
    unsigned hu[128];
    hipLaunchKernelGGL(swap_probe, dim3(1), dim3(64), 0, 0, u);
    hipMemcpy(hu, u, sizeof(hu), hipMemcpyDeviceToHost);
    // expected: r[0] (the first operand): lanes 0-31 keep x, lanes 32-63 receive y of lane l - 32; r[1]: lanes 0-31 receive x of lane l + 32, lanes 32-63 keep y
    int sb = 0;
    for (unsigned l = 0; l < 64; ++l) {
        const unsigned w0 = l < 32 ? 100 + l : 200 + (l - 32), w1 = l < 32 ? 100 + (l + 32) : 200 + l;
        if (hu[2 * l] != w0 || hu[2 * l + 1] != w1) { if (sb < 5) printf("swap lane %u: got (%u, %u), model (%u, %u)\n", l, hu[2 * l], hu[2 * l + 1], w0, w1); ++sb; }
    }
    printf("v_permlane32_swap_b32 (first operand's lanes 32-63 <-> second operand's lanes 0-31): %s\n", sb ? "WRONG MODEL" : "confirmed");
    printf("lane 0: (%u, %u)  lane 31: (%u, %u)  lane 32: (%u, %u)  lane 63: (%u, %u)\n", hu[0], hu[1], hu[62], hu[63], hu[64], hu[65], hu[126], hu[127]);
    return (bad || sb) ? 1 : 0;
}
