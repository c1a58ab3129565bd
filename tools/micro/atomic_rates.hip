// Microbenchmark (gfx950): throughput of global float atomicAdd (no return) issued as 128-B lines (32 lanes = one voxel's
// channels), as the grid-gradient scatter does.  NL distinct lines are hit `reps` times each in total, by waves spread over
// the chip; "window" = how many DIFFERENT lines separate two atomics on the same line in one wave's stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k(float* g, const int* lines, int per_wave, int half_lanes) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int ch = lane & 31, h = lane >> 5;
    for (int i = 0; i < per_wave; ++i) {
        const int l = lines[(long long)wave * per_wave * 2 + 2 * i + h];
        if (half_lanes && h) continue;
        atomicAdd(g + (long long)l * 32 + ch, 1.0f);
    }
}
int main() {
    const int nwaves = 256 * 16, per_wave = 64;
    float* g; int* d;
    const long long nvox = 1 << 20;
    (void)hipMalloc(&g, nvox * 128); (void)hipMemset(g, 0, nvox * 128);
    (void)hipMalloc(&d, (size_t)nwaves * per_wave * 2 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%d waves x %d steps; per step a wave adds to two 128-B lines (or one, 32 lanes)\n", nwaves, per_wave);
    for (int half = 0; half < 2; ++half)
        for (long long nl : {1ll << 20, 1ll << 16, 1ll << 13, 1ll << 10, 1ll << 7, 8ll}) {
            std::vector<int> h((size_t)nwaves * per_wave * 2);
            unsigned s = 12345;
            for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (int)((s >> 8) % nl) * (int)(nvox / nl); }
            (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(nwaves / 4), dim3(256), 0, 0, g, d, per_wave, half);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            const double lines = (double)nwaves * per_wave * (half ? 1 : 2);
            printf("%s  distinct lines %8lld : %8.1f us  %7.2f G line-atomics/s  (%.2f TB/s)\n", half ? "32 lanes" : "64 lanes", nl, best * 1e3,
                   lines / (best * 1e-3) * 1e-9, lines * 128 / (best * 1e-3) * 1e-12);
        }
    return 0;
}
