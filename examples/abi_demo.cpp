// A consumer of libadfp.so that is NOT Python: plain C++ + the HIP runtime + include/adfp.h.
// Reads a scene dumped as raw little-endian arrays (see tests/test_gpu_abi_demo.py for the writer), renders a
// ray batch with adfp_render_forward and writes depth / uncertainty / colour / attention weight back.
//   hipcc -O2 -Iinclude examples/abi_demo.cpp -Lattentive_dfprior_amd -ladfp -o abi_demo
//   LD_LIBRARY_PATH=attentive_dfprior_amd ./abi_demo <dir>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "adfp.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define ADFP_OK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d\n", #x, rc_); return 3; } } while (0)

template <typename T>
static std::vector<T> slurp(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(1); }
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<T> v(n / sizeof(T));
    if (fread(v.data(), 1, n, f) != (size_t)n) { fprintf(stderr, "short read %s\n", path.c_str()); exit(1); }
    fclose(f);
    return v;
}
template <typename T>
static T* upload(const std::vector<T>& v) {
    T* d = nullptr;
    if (hipMalloc(&d, v.size() * sizeof(T)) != hipSuccess) { fprintf(stderr, "hipMalloc\n"); exit(1); }
    if (hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "hipMemcpy\n"); exit(1); }
    return d;
}
template <typename T>
static void dump(const std::string& path, const T* dev, size_t n) {
    std::vector<T> h(n);
    if (hipMemcpy(h.data(), dev, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "hipMemcpy back\n"); exit(1); }
    FILE* f = fopen(path.c_str(), "wb");
    fwrite(h.data(), sizeof(T), n, f);
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <dir>\n", argv[0]); return 1; }
    const std::string d = std::string(argv[1]) + "/";
    // meta: n_rays, n_samples, n_surface, stage, then (Z,Y,X) of low / high / color grids, then TSDF (X,Y,Z physical)
    const std::vector<long long> meta = slurp<long long>(d + "meta.i64");
    const int N = (int)meta[0], NS = (int)meta[1], NF = (int)meta[2], stage = (int)meta[3];
    const std::vector<double> bounds = slurp<double>(d + "bounds.f64");       // bound[3][2], tsdf_bnds[3][2]
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));

    adfp_scene sc = {};
    for (int k = 0; k < 3; ++k) for (int j = 0; j < 2; ++j) { sc.bound[k][j] = bounds[2 * k + j]; sc.tsdf_bnds[k][j] = bounds[6 + 2 * k + j]; }
    const char* names[3] = {"grid_low", "grid_high", "grid_color"};
    adfp_grid* grids[3] = {&sc.low, &sc.high, &sc.color};
    for (int g = 0; g < 3; ++g) {                                              // channel-major [32,Z,Y,X] -> channels-last
        const int Z = (int)meta[4 + 3 * g], Y = (int)meta[5 + 3 * g], X = (int)meta[6 + 3 * g];
        float* cm = upload(slurp<float>(d + names[g] + ".f32"));
        float* cl = nullptr;
        HIP_OK(hipMalloc(&cl, (size_t)32 * Z * Y * X * 4));
        ADFP_OK(adfp_relayout_grid(cm, cl, 32, Z, Y, X, st));
        grids[g]->data = cl; grids[g]->Z = Z; grids[g]->Y = Y; grids[g]->X = X;
    }
    {                                                                          // TSDF stays in its physical [X][Y][Z] order
        const int X = (int)meta[13], Y = (int)meta[14], Z = (int)meta[15];
        sc.tsdf.data = upload(slurp<float>(d + "tsdf_xyz.f32"));
        sc.tsdf.Z = Z; sc.tsdf.Y = Y; sc.tsdf.X = X;
        sc.tsdf.sZ = 1; sc.tsdf.sY = Z; sc.tsdf.sX = (long long)Y * Z;
    }
    HIP_OK(hipMalloc(&sc.status, 4));                                          // sticky status word (f16-range guard)
    HIP_OK(hipMemsetAsync(sc.status, 0, 4, st));
    const char* nets[3] = {"low", "high", "color"};
    const float** w[3] = {&sc.w_low, &sc.w_high, &sc.w_color};
    const void** hw[3] = {&sc.h_low, &sc.h_high, &sc.h_color};
    for (int k = 0; k < 3; ++k) {                                              // flat state_dict-order parameters -> packed images
        float* flat = upload(slurp<float>(d + "flat_" + nets[k] + ".f32"));
        float* packed = nullptr; void* packed_h = nullptr;
        HIP_OK(hipMalloc(&packed, adfp_decoder_packed_floats(k) * 4));
        HIP_OK(hipMalloc(&packed_h, adfp_decoder_packed_h_words(k) * 4));
        ADFP_OK(adfp_pack_decoder(k, flat, packed, st));
        ADFP_OK(adfp_pack_decoder_h(k, flat, packed_h, sc.status, st));
        *w[k] = packed; *hw[k] = packed_h;
    }
    {
        float* flat = upload(slurp<float>(d + "flat_att.f32"));
        float* packed = nullptr; void* packed_h = nullptr;
        HIP_OK(hipMalloc(&packed, adfp_attention_packed_floats() * 4));
        HIP_OK(hipMalloc(&packed_h, adfp_attention_packed_h_words() * 4));
        ADFP_OK(adfp_pack_attention(flat, packed, st));
        ADFP_OK(adfp_pack_attention_h(flat, packed_h, sc.status, st));
        sc.w_att = packed; sc.h_att = packed_h;
    }

    const int S = NS + NF;
    adfp_render_args a = {};
    a.stage = stage; a.n_rays = N; a.n_samples = NS; a.n_surface = NF;
    a.rays_o = upload(slurp<float>(d + "rays_o.f32"));
    a.rays_d = upload(slurp<float>(d + "rays_d.f32"));
    a.gt_depth = upload(slurp<float>(d + "gt_depth.f32"));
    HIP_OK(hipMalloc(&a.depth, (size_t)N * 8));
    HIP_OK(hipMalloc(&a.uncertainty, (size_t)N * 8));
    HIP_OK(hipMalloc(&a.color, (size_t)N * 12));
    HIP_OK(hipMalloc(&a.weight, (size_t)N * S * 4));
    a.workspace_bytes = adfp_workspace_bytes((long long)N * S);
    HIP_OK(hipMalloc(&a.workspace, a.workspace_bytes));
    ADFP_OK(adfp_render_forward(&sc, &a, st));
    HIP_OK(hipStreamSynchronize(st));
    int status = 0;
    HIP_OK(hipMemcpy(&status, sc.status, 4, hipMemcpyDeviceToHost));
    if (status & ADFP_STATUS_F16_RANGE) { fprintf(stderr, "operand beyond the f16 range: use the exact images (h_* = NULL)\n"); return 3; }
    dump(d + "out_depth.f64", a.depth, N);
    dump(d + "out_uncertainty.f64", a.uncertainty, N);
    dump(d + "out_color.f32", a.color, (size_t)N * 3);
    dump(d + "out_weight.f32", a.weight, (size_t)N * S);
    printf("adfp %d: rendered %d rays x %d samples, stage %d\n", adfp_version(), N, S, stage);
    return 0;
}
