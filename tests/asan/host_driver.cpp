// Host-side sanitizer driver ("Race detection / sanitizers": an AddressSanitizer + UBSan build of the C-ABI shim).
// libadfp is rebuilt with -fsanitize=address,undefined on the HOST translation only (-fno-gpu-sanitize: device ASan needs XNACK,
// which this pool does not offer) and this program walks every entry point of include/adfp.h through its host logic:
//   * the argument-error paths (NULL pointers, negative / oversize counts, unknown stages and kinds, missing weight images,
//     workspaces one byte too small) -- each must come back with the NEGATIVE code adfp.h promises, before any launch;
//   * the accepting paths with well-formed descriptors whose device pointers are opaque non-NULL values: the host side carves
//     workspaces, builds job tables and kernel-argument structs and reaches its launches.  Without a GPU (the build container)
//     a launch fails with a positive hipError_t, which is a legal return of the ABI; with a GPU this program is not meant to run.
// A sanitizer report aborts the process (-fno-sanitize-recover); tests/test_asan_host.py builds and runs it in the CPU suite.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "adfp.h"

static int g_fail = 0;
#define EXPECT_NEG(expr) do { const long long rc_ = (long long)(expr); if (rc_ >= 0) { printf("FAIL line %d: %s returned %lld, expected a negative code\n", __LINE__, #expr, rc_); ++g_fail; } } while (0)
#define EXPECT_CODE(expr, code) do { const long long rc_ = (long long)(expr); if (rc_ != (code)) { printf("FAIL line %d: %s returned %lld, expected %d\n", __LINE__, #expr, rc_, (int)(code)); ++g_fail; } } while (0)
// accepted by the host logic: 0 (a GPU executed it -- not expected here) or a positive hipError_t from the first launch
#define EXPECT_REACHES_LAUNCH(expr) do { const long long rc_ = (long long)(expr); if (rc_ < 0) { printf("FAIL line %d: %s returned %lld, expected the host logic to accept it\n", __LINE__, #expr, rc_); ++g_fail; } } while (0)

template <typename T> static T* dev(unsigned long long k) { return (T*)(0x7f0000000000ull + k * 0x10000000ull); }    // opaque "device" addresses

static adfp_scene make_scene() {
    adfp_scene s;
    memset(&s, 0, sizeof(s));
    const double b[3][2] = {{-1.0, 2.0}, {-1.5, 1.5}, {0.0, 2.5}};
    memcpy(s.bound, b, sizeof(b)); memcpy(s.tsdf_bnds, b, sizeof(b));
    s.low.data = dev<float>(1); s.low.Z = 5; s.low.Y = 6; s.low.X = 7;
    s.high.data = dev<float>(2); s.high.Z = 10; s.high.Y = 12; s.high.X = 14;
    s.color.data = dev<float>(3); s.color.Z = 10; s.color.Y = 12; s.color.X = 14;
    s.tsdf.data = dev<float>(4); s.tsdf.Z = 40; s.tsdf.Y = 48; s.tsdf.X = 56; s.tsdf.sZ = 1; s.tsdf.sY = 40; s.tsdf.sX = 40 * 48;
    s.w_low = dev<float>(5); s.w_high = dev<float>(6); s.w_color = dev<float>(7); s.w_att = dev<float>(8);
    return s;
}

int main() {
    void* st = nullptr;
    // ---- sizes and versions
    EXPECT_CODE(adfp_version(), ADFP_VERSION);
    for (int k = 0; k < 3; ++k) {
        if (adfp_decoder_flat_floats(k) <= 0 || adfp_decoder_packed_floats(k) <= 0 || adfp_decoder_packed_h_words(k) <= 0 ||
            adfp_decoder_packed_ht_words(k) <= 0 || adfp_train_act_floats(k) <= 0) { printf("FAIL: size query of decoder kind %d\n", k); ++g_fail; }
    }
    EXPECT_NEG(adfp_decoder_flat_floats(9)); EXPECT_NEG(adfp_decoder_packed_floats(-1)); EXPECT_NEG(adfp_decoder_packed_h_words(3));
    EXPECT_NEG(adfp_decoder_packed_ht_words(7)); EXPECT_NEG(adfp_train_act_floats(5));
    if (adfp_workspace_bytes(0) == 0 || adfp_workspace_bytes(1000) <= adfp_workspace_bytes(10)) { printf("FAIL: workspace bytes\n"); ++g_fail; }
    if (adfp_backward_workspace_bytes(-5) != 0 || adfp_backward_workspace_bytes(64) == 0 || adfp_sort_workspace_bytes(-1) != 0) { printf("FAIL: workspace queries\n"); ++g_fail; }

    // ---- layout / packing
    EXPECT_NEG(adfp_relayout_grid(nullptr, dev<float>(1), 32, 4, 4, 4, st));
    EXPECT_NEG(adfp_relayout_grid(dev<float>(1), dev<float>(2), 32, 0, 4, 4, st));
    EXPECT_NEG(adfp_relayout_grid_back(dev<float>(1), nullptr, 32, 4, 4, 4, st));
    EXPECT_REACHES_LAUNCH(adfp_relayout_grid(dev<float>(1), dev<float>(2), 32, 5, 6, 7, st));
    EXPECT_REACHES_LAUNCH(adfp_relayout_grid_back(dev<float>(1), dev<float>(2), 32, 5, 6, 7, st));
    {   // several grids in one launch
        adfp_relayout_job rj[3] = {{dev<float>(1), dev<float>(2), 210}, {dev<float>(3), dev<float>(4), 1}, {dev<float>(5), dev<float>(6), 4096}};
        EXPECT_CODE(adfp_relayout_grids(0, nullptr, 0, st), 0);
        EXPECT_NEG(adfp_relayout_grids(-1, rj, 0, st));
        EXPECT_NEG(adfp_relayout_grids(ADFP_RELAYOUT_MAX_JOBS + 1, rj, 0, st));
        EXPECT_NEG(adfp_relayout_grids(2, nullptr, 0, st));
        { adfp_relayout_job bad[2] = {rj[0], rj[1]}; bad[1].voxels = 0; EXPECT_NEG(adfp_relayout_grids(2, bad, 0, st)); }
        { adfp_relayout_job bad[2] = {rj[0], rj[1]}; bad[0].dst = nullptr; EXPECT_NEG(adfp_relayout_grids(2, bad, 1, st)); }
        EXPECT_REACHES_LAUNCH(adfp_relayout_grids(3, rj, 0, st));
        EXPECT_REACHES_LAUNCH(adfp_relayout_grids(3, rj, 1, st));
    }
    for (int k = -1; k < 4; ++k) {
        if (k < 0 || k > 2) {
            EXPECT_NEG(adfp_pack_decoder(k, dev<float>(1), dev<float>(2), st));
            EXPECT_NEG(adfp_pack_decoder_h(k, dev<float>(1), dev<void>(2), nullptr, st));
            EXPECT_NEG(adfp_pack_decoder_ht(k, dev<float>(1), dev<void>(2), nullptr, st));
        } else {
            EXPECT_REACHES_LAUNCH(adfp_pack_decoder(k, dev<float>(1), dev<float>(2), st));
            EXPECT_REACHES_LAUNCH(adfp_pack_decoder_h(k, dev<float>(1), dev<void>(2), dev<int>(3), st));
            EXPECT_REACHES_LAUNCH(adfp_pack_decoder_ht(k, dev<float>(1), dev<void>(2), nullptr, st));
            EXPECT_NEG(adfp_pack_decoder(k, nullptr, dev<float>(2), st));
            EXPECT_NEG(adfp_pack_decoder_h(k, dev<float>(1), nullptr, nullptr, st));
        }
    }
    EXPECT_NEG(adfp_pack_split_image(4, ADFP_IMAGE_H, dev<float>(1), dev<void>(2), nullptr, st));
    EXPECT_NEG(adfp_pack_split_image(ADFP_NET_ATT, 0, dev<float>(1), dev<void>(2), nullptr, st));
    EXPECT_NEG(adfp_pack_split_image(ADFP_NET_ATT, 4, dev<float>(1), dev<void>(2), nullptr, st));
    for (int net = 0; net < 4; ++net) for (int which = 1; which < 4; ++which) EXPECT_REACHES_LAUNCH(adfp_pack_split_image(net, which, dev<float>(1), dev<void>(2), dev<int>(3), st));
    EXPECT_NEG(adfp_pack_attention(nullptr, dev<float>(1), st)); EXPECT_NEG(adfp_pack_attention_h(dev<float>(1), nullptr, nullptr, st));
    EXPECT_NEG(adfp_pack_attention_ht(nullptr, nullptr, nullptr, st));
    EXPECT_REACHES_LAUNCH(adfp_pack_attention(dev<float>(1), dev<float>(2), st));
    EXPECT_REACHES_LAUNCH(adfp_pack_attention_h(dev<float>(1), dev<void>(2), nullptr, st));
    EXPECT_REACHES_LAUNCH(adfp_pack_attention_ht(dev<float>(1), dev<void>(2), dev<int>(3), st));

    // ---- rays, pre-filter, sampler
    EXPECT_NEG(adfp_get_rays(0, 10, 1.f, 1.f, 0.f, 0.f, dev<float>(1), dev<float>(2), dev<float>(3), st));
    EXPECT_NEG(adfp_get_rays(10, 10, 1.f, 1.f, 0.f, 0.f, nullptr, dev<float>(2), dev<float>(3), st));
    EXPECT_REACHES_LAUNCH(adfp_get_rays(48, 64, 60.f, 60.f, 31.5f, 23.5f, dev<float>(1), dev<float>(2), dev<float>(3), st));
    EXPECT_NEG(adfp_rays_from_uv(nullptr, dev<float>(1), 10, 1, 1, 0, 0, dev<float>(2), dev<float>(3), dev<float>(4), st));
    EXPECT_NEG(adfp_rays_from_uv(dev<float>(1), dev<float>(1), -1, 1, 1, 0, 0, dev<float>(2), dev<float>(3), dev<float>(4), st));
    EXPECT_REACHES_LAUNCH(adfp_rays_from_uv(dev<float>(1), dev<float>(5), 1000, 60, 60, 31.5f, 23.5f, dev<float>(2), dev<float>(3), dev<float>(4), st));
    EXPECT_NEG(adfp_rays_from_uv_backward(dev<float>(1), dev<float>(5), 10, 60, 60, 0, 0, dev<float>(2), dev<float>(3), nullptr, st));
    EXPECT_REACHES_LAUNCH(adfp_rays_from_uv_backward(dev<float>(1), dev<float>(5), 10, 60, 60, 0, 0, dev<float>(2), nullptr, dev<float>(4), st));
    EXPECT_NEG(adfp_prefilter_rays(dev<float>(1), dev<float>(2), nullptr, 10, dev<double>(3), dev<int>(4), dev<int>(5), st));
    EXPECT_NEG(adfp_prefilter_rays(dev<float>(1), dev<float>(2), dev<float>(6), -3, dev<double>(3), dev<int>(4), dev<int>(5), st));
    EXPECT_REACHES_LAUNCH(adfp_prefilter_rays(dev<float>(1), dev<float>(2), dev<float>(6), 5000, dev<double>(3), dev<int>(4), dev<int>(5), st));
    EXPECT_NEG(adfp_prefilter_mask(dev<float>(1), dev<float>(2), dev<float>(6), 10, nullptr, dev<unsigned char>(4), dev<float>(5), st));
    EXPECT_REACHES_LAUNCH(adfp_prefilter_mask(dev<float>(1), dev<float>(2), dev<float>(6), 10, dev<double>(3), dev<unsigned char>(4), dev<float>(5), st));
    const double bound[3][2] = {{-1, 2}, {-1.5, 1.5}, {0, 2.5}};
    EXPECT_NEG(adfp_sample_rays(dev<float>(1), dev<float>(2), dev<float>(3), 100, bound, 0, 16, 0, 0.f, nullptr, nullptr, dev<double>(4), dev<void>(5), st));
    EXPECT_CODE(adfp_sample_rays(dev<float>(1), dev<float>(2), dev<float>(3), 100, bound, 250, 16, 0, 0.f, nullptr, nullptr, dev<double>(4), dev<void>(5), st), ADFP_E_UNSUPPORTED);
    EXPECT_NEG(adfp_sample_rays(dev<float>(1), dev<float>(2), dev<float>(3), 100, bound, 32, 16, 0, 0.5f, nullptr, nullptr, dev<double>(4), dev<void>(5), st));   // perturb without t_rand
    EXPECT_REACHES_LAUNCH(adfp_sample_rays(dev<float>(1), dev<float>(2), dev<float>(3), 100, bound, 32, 16, 0, 0.f, nullptr, nullptr, dev<double>(4), dev<void>(5), st));
    EXPECT_REACHES_LAUNCH(adfp_sample_rays(dev<float>(1), dev<float>(2), nullptr, 100, bound, 32, 16, 1, 0.f, nullptr, dev<float>(6), dev<double>(4), dev<void>(5), st));

    // ---- point queries
    adfp_scene sc = make_scene();
    adfp_points pts; memset(&pts, 0, sizeof(pts));
    pts.mode = ADFP_PTS_F64; pts.n_points = 1000; pts.pts = dev<double>(9);
    const size_t need = adfp_workspace_bytes(1000);
    EXPECT_NEG(adfp_eval_points(nullptr, &pts, ADFP_STAGE_COLOR, 0, dev<float>(10), dev<float>(11), dev<void>(12), need, st));
    EXPECT_NEG(adfp_eval_points(&sc, nullptr, ADFP_STAGE_COLOR, 0, dev<float>(10), dev<float>(11), dev<void>(12), need, st));
    EXPECT_NEG(adfp_eval_points(&sc, &pts, 7, 0, dev<float>(10), dev<float>(11), dev<void>(12), need, st));
    EXPECT_NEG(adfp_eval_points(&sc, &pts, ADFP_STAGE_COLOR, 0, nullptr, dev<float>(11), dev<void>(12), need, st));
    EXPECT_CODE(adfp_eval_points(&sc, &pts, ADFP_STAGE_COLOR, 0, dev<float>(10), dev<float>(11), dev<void>(12), need - 1, st), ADFP_E_WORKSPACE);
    { adfp_scene s2 = sc; s2.w_color = nullptr; EXPECT_NEG(adfp_eval_points(&s2, &pts, ADFP_STAGE_COLOR, 0, dev<float>(10), dev<float>(11), dev<void>(12), need, st));
      EXPECT_REACHES_LAUNCH(adfp_eval_points(&s2, &pts, ADFP_STAGE_HIGH, 0, dev<float>(10), dev<float>(11), dev<void>(12), need, st)); }
    { adfp_scene s2 = sc; s2.tsdf.data = nullptr; EXPECT_NEG(adfp_eval_points(&s2, &pts, ADFP_STAGE_HIGH, 0, dev<float>(10), dev<float>(11), dev<void>(12), need, st));
      EXPECT_REACHES_LAUNCH(adfp_eval_points(&s2, &pts, ADFP_STAGE_LOW, 0, dev<float>(10), dev<float>(11), dev<void>(12), need, st)); }
    { adfp_scene s2 = sc; s2.high.Z = 4096; s2.high.Y = 4096; s2.high.X = 4096; EXPECT_CODE(adfp_eval_points(&s2, &pts, ADFP_STAGE_COLOR, 0, dev<float>(10), dev<float>(11), dev<void>(12), need, st), ADFP_E_UNSUPPORTED); }
    { adfp_points p2 = pts; p2.n_points = -1; EXPECT_NEG(adfp_eval_points(&sc, &p2, ADFP_STAGE_COLOR, 0, dev<float>(10), dev<float>(11), dev<void>(12), need, st));
      p2.n_points = 0x100000000ll; EXPECT_NEG(adfp_eval_points(&sc, &p2, ADFP_STAGE_COLOR, 0, dev<float>(10), dev<float>(11), dev<void>(12), (size_t)1 << 40, st));
      p2 = pts; p2.mode = 17; EXPECT_NEG(adfp_eval_points(&sc, &p2, ADFP_STAGE_COLOR, 0, dev<float>(10), dev<float>(11), dev<void>(12), need, st));
      p2 = pts; p2.mode = ADFP_PTS_RAYS; EXPECT_NEG(adfp_eval_points(&sc, &p2, ADFP_STAGE_COLOR, 0, dev<float>(10), dev<float>(11), dev<void>(12), need, st));     // ray mode without rays
      p2 = pts; p2.n_points = 0; EXPECT_CODE(adfp_eval_points(&sc, &p2, ADFP_STAGE_COLOR, 0, dev<float>(10), dev<float>(11), dev<void>(12), need, st), 0); }
    EXPECT_REACHES_LAUNCH(adfp_eval_points(&sc, &pts, ADFP_STAGE_COLOR, 1, dev<float>(10), dev<float>(11), dev<void>(12), need, st));
    { adfp_scene s2 = sc; s2.h_low = dev<void>(20); s2.h_color = dev<void>(21); s2.h_high = dev<void>(22); s2.h_att = dev<void>(23);      // the f16-split images + repair path
      s2.flat_low = dev<float>(24); s2.flat_high = dev<float>(25); s2.flat_color = dev<float>(26); s2.flat_att = dev<float>(27); s2.status = dev<int>(28);
      for (int stage = 0; stage < 3; ++stage) EXPECT_REACHES_LAUNCH(adfp_eval_points(&s2, &pts, stage, 1, dev<float>(10), dev<float>(11), dev<void>(12), need, st)); }
    adfp_train_state ts; memset(&ts, 0, sizeof(ts));
    EXPECT_NEG(adfp_eval_points_train(&sc, &pts, ADFP_STAGE_COLOR, 0, dev<float>(10), dev<float>(11), dev<void>(12), need, &ts, st));      // training state without its buffers
    ts.flags = dev<unsigned char>(30); ts.list = dev<int>(31); ts.counter = dev<int>(32); ts.att_occ = dev<float>(33); ts.att_u = dev<float>(34);
    EXPECT_REACHES_LAUNCH(adfp_eval_points_train(&sc, &pts, ADFP_STAGE_COLOR, 0, dev<float>(10), dev<float>(11), dev<void>(12), need, &ts, st));
    const double tb[3][2] = {{-1, 2}, {-1.5, 1.5}, {0, 2.5}};
    EXPECT_NEG(adfp_sample_tsdf(nullptr, tb, &pts, dev<float>(10), st));
    EXPECT_NEG(adfp_sample_tsdf(&sc.tsdf, tb, &pts, nullptr, st));
    EXPECT_REACHES_LAUNCH(adfp_sample_tsdf(&sc.tsdf, tb, &pts, dev<float>(10), st));
    EXPECT_NEG(adfp_tsdf_stage(&sc, &pts, dev<unsigned char>(4), dev<int>(1), nullptr, nullptr, dev<int>(3), st));       // a list without its att_u
    EXPECT_REACHES_LAUNCH(adfp_tsdf_stage(&sc, &pts, nullptr, nullptr, nullptr, dev<float>(2), nullptr, st));               // flags / list are optional
    EXPECT_REACHES_LAUNCH(adfp_tsdf_stage(&sc, &pts, dev<unsigned char>(4), dev<int>(1), dev<float>(2), nullptr, dev<int>(3), st));
    EXPECT_CODE(adfp_decode_stage(&sc, &pts, 5, dev<float>(10), dev<float>(11), nullptr, st), ADFP_E_UNSUPPORTED);
    EXPECT_NEG(adfp_decode_stage(&sc, &pts, ADFP_DEC_LOW_COLOR, dev<float>(10), dev<float>(11), nullptr, st));       // the fused launch needs the split images
    EXPECT_REACHES_LAUNCH(adfp_decode_stage(&sc, &pts, ADFP_DEC_LOW, dev<float>(10), dev<float>(11), nullptr, st));
    EXPECT_NEG(adfp_decode_single(&sc, &pts, 4, dev<float>(10), st));
    EXPECT_NEG(adfp_decode_single(&sc, &pts, ADFP_DEC_HIGH, nullptr, st));
    for (int k = 0; k < 3; ++k) EXPECT_REACHES_LAUNCH(adfp_decode_single(&sc, &pts, k, dev<float>(10), st));
    EXPECT_NEG(adfp_attention_rows(&sc, nullptr, dev<float>(1), 100, dev<float>(2), dev<float>(3), dev<float>(4), st));
    EXPECT_NEG(adfp_attention_rows(&sc, dev<float>(5), dev<float>(1), -1, dev<float>(2), dev<float>(3), dev<float>(4), st));
    EXPECT_REACHES_LAUNCH(adfp_attention_rows(&sc, dev<float>(5), dev<float>(1), 100, dev<float>(2), dev<float>(3), dev<float>(4), st));

    // ---- compositing, render forward / backward
    EXPECT_NEG(adfp_composite(dev<float>(1), dev<double>(2), 10, 0, dev<double>(3), dev<double>(4), dev<float>(5), nullptr, st));
    EXPECT_NEG(adfp_composite(nullptr, dev<double>(2), 10, 48, dev<double>(3), dev<double>(4), dev<float>(5), nullptr, st));
    EXPECT_REACHES_LAUNCH(adfp_composite(dev<float>(1), dev<double>(2), 10, 48, dev<double>(3), dev<double>(4), dev<float>(5), nullptr, st));
    adfp_render_args ra; memset(&ra, 0, sizeof(ra));
    ra.stage = ADFP_STAGE_COLOR; ra.n_rays = 500; ra.n_samples = 32; ra.n_surface = 16;
    ra.rays_o = dev<float>(1); ra.rays_d = dev<float>(2); ra.gt_depth = dev<float>(3);
    ra.depth = dev<double>(4); ra.uncertainty = dev<double>(5); ra.color = dev<float>(6); ra.weight = dev<float>(7);
    ra.workspace = dev<void>(8); ra.workspace_bytes = adfp_workspace_bytes(500 * 48);
    EXPECT_NEG(adfp_render_forward(&sc, nullptr, st));
    EXPECT_NEG(adfp_render_forward(nullptr, &ra, st));
    { adfp_render_args r2 = ra; r2.workspace_bytes -= 1; EXPECT_CODE(adfp_render_forward(&sc, &r2, st), ADFP_E_WORKSPACE);
      r2 = ra; r2.color = nullptr; EXPECT_NEG(adfp_render_forward(&sc, &r2, st));
      r2 = ra; r2.n_samples = 250; r2.workspace_bytes = (size_t)1 << 40; EXPECT_CODE(adfp_render_forward(&sc, &r2, st), ADFP_E_UNSUPPORTED);
      r2 = ra; r2.n_rays = -2; EXPECT_NEG(adfp_render_forward(&sc, &r2, st));
      r2 = ra; r2.depth_max_segment = -1; EXPECT_NEG(adfp_render_forward(&sc, &r2, st));
      r2 = ra; r2.stage = 3; EXPECT_NEG(adfp_render_forward(&sc, &r2, st));
      r2 = ra; r2.n_rays = 0; EXPECT_CODE(adfp_render_forward(&sc, &r2, st), 0);
      r2 = ra; r2.state = &ts; EXPECT_REACHES_LAUNCH(adfp_render_forward(&sc, &r2, st));
      {   // weight images handed over to the call's first launch
          adfp_pack_job pj[2] = {{ADFP_DEC_COLOR, ADFP_IMAGE_H, dev<float>(31), dev<float>(32)}, {ADFP_NET_ATT, ADFP_IMAGE_HT, dev<float>(33), dev<float>(34)}};
          r2 = ra; r2.pack_jobs = pj; r2.n_pack_jobs = -1; EXPECT_NEG(adfp_render_forward(&sc, &r2, st));
          r2 = ra; r2.pack_jobs = pj; r2.n_pack_jobs = ADFP_PACK_MAX_JOBS + 1; EXPECT_NEG(adfp_render_forward(&sc, &r2, st));
          r2 = ra; r2.pack_jobs = nullptr; r2.n_pack_jobs = 2; EXPECT_NEG(adfp_render_forward(&sc, &r2, st));
          r2 = ra; r2.pack_jobs = pj; r2.n_pack_jobs = 2; pj[1].format = 99; EXPECT_NEG(adfp_render_forward(&sc, &r2, st)); pj[1].format = ADFP_IMAGE_HT;
          r2 = ra; r2.pack_jobs = pj; r2.n_pack_jobs = 2; EXPECT_REACHES_LAUNCH(adfp_render_forward(&sc, &r2, st));
          r2 = ra; r2.pack_jobs = pj; r2.n_pack_jobs = 2; r2.n_rays = 0; EXPECT_REACHES_LAUNCH(adfp_render_forward(&sc, &r2, st));     // no rays: the images are still packed
      }
      {   // grids handed over to the call's first launch
          adfp_relayout_job rj[2] = {{dev<float>(35), dev<float>(36), 210}, {dev<float>(37), dev<float>(38), 4096}};
          r2 = ra; r2.relayout_jobs = rj; r2.n_relayout_jobs = -1; EXPECT_NEG(adfp_render_forward(&sc, &r2, st));
          r2 = ra; r2.relayout_jobs = nullptr; r2.n_relayout_jobs = 1; EXPECT_NEG(adfp_render_forward(&sc, &r2, st));
          r2 = ra; r2.relayout_jobs = rj; r2.n_relayout_jobs = ADFP_RELAYOUT_MAX_JOBS + 1; EXPECT_NEG(adfp_render_forward(&sc, &r2, st));
          r2 = ra; r2.relayout_jobs = rj; r2.n_relayout_jobs = 2; EXPECT_REACHES_LAUNCH(adfp_render_forward(&sc, &r2, st));
          r2 = ra; r2.relayout_jobs = rj; r2.n_relayout_jobs = 2; r2.n_rays = 0; EXPECT_REACHES_LAUNCH(adfp_render_forward(&sc, &r2, st));      // no rays: still converted
      }
      {   // the pre-filter as a job of the first launch
          r2 = ra; r2.prefilter_bound = dev<double>(39); EXPECT_NEG(adfp_render_forward(&sc, &r2, st));                       // bound without keep
          r2 = ra; r2.prefilter_keep = dev<unsigned char>(40); EXPECT_NEG(adfp_render_forward(&sc, &r2, st));                 // keep without bound
          r2 = ra; r2.prefilter_bound = dev<double>(39); r2.prefilter_keep = dev<unsigned char>(40); r2.depth_max = dev<float>(21); EXPECT_NEG(adfp_render_forward(&sc, &r2, st));
          r2 = ra; r2.prefilter_bound = dev<double>(39); r2.prefilter_keep = dev<unsigned char>(40); r2.depth_max_segment = 100; EXPECT_NEG(adfp_render_forward(&sc, &r2, st));
          r2 = ra; r2.prefilter_bound = dev<double>(39); r2.prefilter_keep = dev<unsigned char>(40); r2.gt_depth = nullptr; EXPECT_NEG(adfp_render_forward(&sc, &r2, st));
          r2 = ra; r2.prefilter_bound = dev<double>(39); r2.prefilter_keep = dev<unsigned char>(40); EXPECT_REACHES_LAUNCH(adfp_render_forward(&sc, &r2, st));
      }
      adfp_train_state t0; memset(&t0, 0, sizeof(t0)); r2.state = &t0; EXPECT_NEG(adfp_render_forward(&sc, &r2, st));
      r2 = ra; r2.n_rays = 60000000; r2.workspace_bytes = (size_t)1 << 44; EXPECT_CODE(adfp_render_forward(&sc, &r2, st), ADFP_E_UNSUPPORTED); }     // 2.9e9 points
    EXPECT_REACHES_LAUNCH(adfp_render_forward(&sc, &ra, st));
    { adfp_render_args r2 = ra; r2.depth_max_segment = 100; r2.n_rays = 4800; r2.workspace_bytes = adfp_workspace_bytes(4800 * 48); EXPECT_REACHES_LAUNCH(adfp_render_forward(&sc, &r2, st));
      r2.depth_max_segment = 10; EXPECT_CODE(adfp_render_forward(&sc, &r2, st), ADFP_E_UNSUPPORTED);                                                        // 480 segments > 48
      // a ray shard of a segmented frame: needs the frame's maxima from the caller
      r2.depth_max_segment = 100; r2.depth_max_first_ray = 250; EXPECT_NEG(adfp_render_forward(&sc, &r2, st));
      r2.depth_max = dev<float>(21); EXPECT_REACHES_LAUNCH(adfp_render_forward(&sc, &r2, st));
      r2.depth_max_first_ray = -1; EXPECT_NEG(adfp_render_forward(&sc, &r2, st));
      r2.depth_max_first_ray = 250; r2.depth_max_segment = 0; EXPECT_NEG(adfp_render_forward(&sc, &r2, st)); }
    {   // a frame job: the call's rays are pixels [first, first + n_rays) of an H x W frame, rays and far clamps from the first launch
        adfp_frame_job fj; memset(&fj, 0, sizeof(fj));
        fj.c2w = dev<float>(41); fj.H = 60; fj.W = 80; fj.fx = fj.fy = 50.f; fj.cx = 39.5f; fj.cy = 29.5f;
        fj.depth = dev<float>(42); fj.rays_o = dev<float>(43); fj.rays_d = dev<float>(44);
        adfp_render_args r2 = ra; r2.rays_o = r2.rays_d = r2.gt_depth = nullptr; r2.frame = &fj;
        r2.n_rays = 1200; r2.depth_max_segment = 1000; r2.depth_max_first_ray = 2400; r2.workspace_bytes = adfp_workspace_bytes(1200 * 48);
        EXPECT_REACHES_LAUNCH(adfp_render_forward(&sc, &r2, st));
        { adfp_render_args r3 = r2; r3.depth_max_segment = 0; EXPECT_NEG(adfp_render_forward(&sc, &r3, st)); }            // a frame is segmented
        { adfp_render_args r3 = r2; r3.depth_max = dev<float>(21); EXPECT_NEG(adfp_render_forward(&sc, &r3, st)); }       // the maxima come from the frame
        { adfp_render_args r3 = r2; r3.depth_max_first_ray = 4000; EXPECT_NEG(adfp_render_forward(&sc, &r3, st)); }       // 4000 + 1200 > 60 x 80
        { adfp_render_args r3 = r2; r3.depth_max_segment = 10; EXPECT_CODE(adfp_render_forward(&sc, &r3, st), ADFP_E_UNSUPPORTED); }   // 480 segments of the FRAME
        { adfp_render_args r3 = r2; r3.perturb = 0.5f; r3.t_rand = dev<float>(45); EXPECT_CODE(adfp_render_forward(&sc, &r3, st), ADFP_E_UNSUPPORTED); }
        { adfp_frame_job f2 = fj; f2.rays_d = nullptr; adfp_render_args r3 = r2; r3.frame = &f2; EXPECT_NEG(adfp_render_forward(&sc, &r3, st)); }
        { adfp_frame_job f2 = fj; f2.depth = nullptr; adfp_render_args r3 = r2; r3.frame = &f2; EXPECT_NEG(adfp_render_forward(&sc, &r3, st)); }
        { adfp_frame_job f2 = fj; f2.W = 0; adfp_render_args r3 = r2; r3.frame = &f2; EXPECT_NEG(adfp_render_forward(&sc, &r3, st)); }
    }
    adfp_backward_args ba; memset(&ba, 0, sizeof(ba));
    ba.stage = ADFP_STAGE_COLOR; ba.n_rays = 500; ba.S = 48; ba.rays_o = dev<float>(1); ba.rays_d = dev<float>(2); ba.z_vals = dev<double>(3); ba.raw = dev<float>(4);
    ba.state = ts; ba.g_depth = dev<double>(5); ba.g_color = dev<float>(6);
    ba.g_grid_low = dev<float>(7); ba.g_grid_high = dev<float>(8); ba.g_grid_color = dev<float>(9);
    ba.g_flat_color = dev<float>(10); ba.g_flat_att = dev<float>(11); ba.g_flat_low = dev<float>(12); ba.g_flat_high = dev<float>(13);
    ba.workspace = dev<void>(14); ba.workspace_bytes = adfp_backward_workspace_bytes(500 * 48);
    EXPECT_NEG(adfp_render_backward(&sc, nullptr, st));
    { adfp_backward_args b2 = ba; b2.workspace_bytes -= 1; EXPECT_CODE(adfp_render_backward(&sc, &b2, st), ADFP_E_WORKSPACE);
      b2 = ba; b2.raw = nullptr; EXPECT_NEG(adfp_render_backward(&sc, &b2, st));
      b2 = ba; b2.S = 0; EXPECT_NEG(adfp_render_backward(&sc, &b2, st));
      b2 = ba; b2.S = 1 << 20; b2.workspace_bytes = (size_t)1 << 44; EXPECT_CODE(adfp_render_backward(&sc, &b2, st), ADFP_E_UNSUPPORTED);
      b2 = ba; memset(&b2.state, 0, sizeof(b2.state)); EXPECT_NEG(adfp_render_backward(&sc, &b2, st));
      b2 = ba; b2.g_rays_o = dev<float>(15); b2.g_rays_d = dev<float>(16); EXPECT_REACHES_LAUNCH(adfp_render_backward(&sc, &b2, st));
      // the side lane: a second stream needs its two events, and must not be the call's own stream
      b2 = ba; b2.side_stream = dev<void>(51); EXPECT_NEG(adfp_render_backward(&sc, &b2, st));
      b2 = ba; b2.side_stream = dev<void>(51); b2.side_events[0] = dev<void>(52); EXPECT_NEG(adfp_render_backward(&sc, &b2, st));
      b2 = ba; b2.side_stream = dev<void>(54); b2.side_events[0] = dev<void>(52); b2.side_events[1] = dev<void>(53); EXPECT_NEG(adfp_render_backward(&sc, &b2, dev<void>(54)));
      b2 = ba; b2.options = ADFP_BWD_SCATTER_IN_KERNEL | ADFP_BWD_GRIDS_PREZEROED | ADFP_BWD_STAGED_WGRAD; EXPECT_REACHES_LAUNCH(adfp_render_backward(&sc, &b2, st));
      adfp_scene s2 = sc; s2.w_att = nullptr; EXPECT_NEG(adfp_render_backward(&s2, &ba, st)); }
    EXPECT_REACHES_LAUNCH(adfp_render_backward(&sc, &ba, st));
    { adfp_scene s2 = sc; s2.ht_low = dev<void>(40); s2.ht_high = dev<void>(41); s2.ht_color = dev<void>(42); s2.ht_att = dev<void>(43); s2.status = dev<int>(44);
      adfp_backward_args b2 = ba;                                         // the f16-split backward with the forward's masks and layer inputs
      b2.state.masks_low = dev<unsigned>(45); b2.state.masks_high = dev<unsigned>(46); b2.state.masks_color = dev<unsigned>(47); b2.state.masks_att = dev<unsigned>(48);
      b2.state.act_low = dev<float>(49); b2.state.act_high = dev<float>(50); b2.state.act_color = dev<float>(51); b2.state.act_att = dev<float>(52);
      for (int stage = 0; stage < 3; ++stage) { b2.stage = stage; EXPECT_REACHES_LAUNCH(adfp_render_backward(&s2, &b2, st)); } }
    adfp_points_backward_args pb; memset(&pb, 0, sizeof(pb));
    pb.stage = ADFP_STAGE_COLOR; pb.state = ts; pb.g_raw = dev<float>(1); pb.g_grid_low = dev<float>(2); pb.g_flat_color = dev<float>(3); pb.g_pts = dev<float>(4);
    pb.workspace = dev<void>(5); pb.workspace_bytes = adfp_backward_workspace_bytes(1000);
    EXPECT_NEG(adfp_eval_points_backward(&sc, &pts, nullptr, st));
    { adfp_points_backward_args p2 = pb; p2.workspace_bytes -= 1; EXPECT_CODE(adfp_eval_points_backward(&sc, &pts, &p2, st), ADFP_E_WORKSPACE);
      p2 = pb; p2.workspace = nullptr; EXPECT_NEG(adfp_eval_points_backward(&sc, &pts, &p2, st)); }
    EXPECT_REACHES_LAUNCH(adfp_eval_points_backward(&sc, &pts, &pb, st));

    // ---- mapping / tracking helpers
    const float c2w[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    EXPECT_NEG(adfp_frustum_mask(0, 6, 5, bound, c2w, c2w, 60, 60, 31.5, 23.5, 48, 64, dev<float>(1), dev<float>(2), dev<unsigned>(3), dev<unsigned char>(4), st));
    EXPECT_NEG(adfp_frustum_mask(7, 6, 5, bound, c2w, c2w, 60, 60, 31.5, 23.5, 48, 64, nullptr, dev<float>(2), dev<unsigned>(3), dev<unsigned char>(4), st));
    EXPECT_REACHES_LAUNCH(adfp_frustum_mask(7, 6, 5, bound, c2w, c2w, 60, 60, 31.5, 23.5, 48, 64, dev<float>(1), dev<float>(2), dev<unsigned>(3), dev<unsigned char>(4), st));
    EXPECT_NEG(adfp_masked_adam(nullptr, dev<float>(1), dev<float>(2), dev<float>(3), nullptr, 100, 32, 1e-3f, 0.9f, 0.999f, 1e-8f, 1, st));
    EXPECT_NEG(adfp_masked_adam(dev<float>(4), dev<float>(1), dev<float>(2), dev<float>(3), nullptr, 100, 32, 1e-3f, 0.9f, 0.999f, 1e-8f, 0, st));     // step counts from 1
    EXPECT_REACHES_LAUNCH(adfp_masked_adam(dev<float>(4), dev<float>(1), dev<float>(2), dev<float>(3), dev<unsigned char>(5), 100, 32, 1e-3f, 0.9f, 0.999f, 1e-8f, 3, st));
    const float lrs[3] = {1e-3f, -1.f, 0.f};
    EXPECT_NEG(adfp_adam_prep(nullptr, dev<float>(1), 3, lrs, 0.9f, 0.999f, nullptr, st));
    EXPECT_NEG(adfp_adam_prep(dev<int>(2), dev<float>(1), 0, lrs, 0.9f, 0.999f, nullptr, st));
    EXPECT_REACHES_LAUNCH(adfp_adam_prep(dev<int>(2), dev<float>(1), 3, lrs, 0.9f, 0.999f, dev<int>(3), st));
    EXPECT_NEG(adfp_masked_adam_dev(dev<float>(4), nullptr, dev<float>(2), dev<float>(3), nullptr, 100, 32, 0.9f, 0.999f, 1e-8f, dev<float>(5), st));
    EXPECT_REACHES_LAUNCH(adfp_masked_adam_dev(dev<float>(4), dev<float>(1), dev<float>(2), dev<float>(3), nullptr, 100, 32, 0.9f, 0.999f, 1e-8f, dev<float>(5), st));
    adfp_adam_group groups[3]; memset(groups, 0, sizeof(groups));
    for (int k = 0; k < 3; ++k) { groups[k].param = dev<float>(1 + k); groups[k].grad = dev<float>(5 + k); groups[k].exp_avg = dev<float>(9 + k); groups[k].exp_avg_sq = dev<float>(13 + k);
                                  groups[k].nvox = 1000 * (k + 1); groups[k].channels = k == 2 ? 32 : 1; groups[k].derived = dev<float>(17 + k); }
    EXPECT_CODE(adfp_masked_adam_multi(0, groups, 0.9f, 0.999f, 1e-8f, st), 0);          // no group: nothing to do
    EXPECT_NEG(adfp_masked_adam_multi(3, nullptr, 0.9f, 0.999f, 1e-8f, st));
    EXPECT_NEG(adfp_masked_adam_multi(99, groups, 0.9f, 0.999f, 1e-8f, st));
    { adfp_adam_group g2[3]; memcpy(g2, groups, sizeof(groups)); g2[1].grad = nullptr; EXPECT_NEG(adfp_masked_adam_multi(3, g2, 0.9f, 0.999f, 1e-8f, st)); }
    EXPECT_REACHES_LAUNCH(adfp_masked_adam_multi(3, groups, 0.9f, 0.999f, 1e-8f, st));
    adfp_adam_cl_group cg[3]; memset(cg, 0, sizeof(cg));
    for (int k = 0; k < 3; ++k) { cg[k].param_cl = dev<float>(1 + k); cg[k].param_cm = dev<float>(4 + k); cg[k].grad_cl = dev<float>(7 + k); cg[k].exp_avg_cl = dev<float>(10 + k);
                                  cg[k].exp_avg_sq_cl = dev<float>(13 + k); cg[k].nvox = 5 * 6 * 7 * (k + 1); cg[k].derived = dev<float>(16 + k); }
    EXPECT_CODE(adfp_adam_grids_cl(0, cg, 0.9f, 0.999f, 1e-8f, st), 0);
    EXPECT_NEG(adfp_adam_grids_cl(99, cg, 0.9f, 0.999f, 1e-8f, st));
    EXPECT_REACHES_LAUNCH(adfp_adam_grids_cl(3, cg, 0.9f, 0.999f, 1e-8f, st));
    EXPECT_CODE(adfp_adam_step(0, cg, 0, groups, 0.9f, 0.999f, 1e-8f, st), 0);              // nothing to step
    EXPECT_NEG(adfp_adam_step(99, cg, 3, groups, 0.9f, 0.999f, 1e-8f, st));
    EXPECT_NEG(adfp_adam_step(3, cg, 99, groups, 0.9f, 0.999f, 1e-8f, st));
    EXPECT_NEG(adfp_adam_step(3, nullptr, 3, groups, 0.9f, 0.999f, 1e-8f, st));
    EXPECT_NEG(adfp_adam_step(3, cg, 3, nullptr, 0.9f, 0.999f, 1e-8f, st));
    EXPECT_REACHES_LAUNCH(adfp_adam_step(3, cg, 0, nullptr, 0.9f, 0.999f, 1e-8f, st));
    EXPECT_REACHES_LAUNCH(adfp_adam_step(0, nullptr, 3, groups, 0.9f, 0.999f, 1e-8f, st));
    EXPECT_REACHES_LAUNCH(adfp_adam_step(3, cg, 3, groups, 0.9f, 0.999f, 1e-8f, st));
    adfp_loss_args la; memset(&la, 0, sizeof(la));
    la.n_rays = 500; la.S = 48; la.stage = ADFP_STAGE_COLOR; la.w_color_loss = 0.2f; la.depth = dev<double>(1); la.color = dev<float>(2); la.weight = dev<float>(3);
    la.gt_depth = dev<float>(4); la.gt_color = dev<float>(5); la.loss = dev<double>(6); la.g_depth = dev<double>(7); la.g_color = dev<float>(8);
    EXPECT_NEG(adfp_mapper_loss(nullptr, st));
    { adfp_loss_args l2 = la; l2.loss = nullptr; EXPECT_NEG(adfp_mapper_loss(&l2, st)); l2 = la; l2.warmup = 1; EXPECT_NEG(adfp_mapper_loss(&l2, st)); }    // warm-up needs g_weight
    EXPECT_REACHES_LAUNCH(adfp_mapper_loss(&la, st));
    {   // the three-in-one form: scratch must be there, aligned and large enough; at most 8 Adam groups, with their arrays
        float lrs[8] = {0.1f, -1.f, 0.005f, 0.005f, 0.005f, 0.f, 0.f, 0.f};
        const size_t sb = adfp_mapper_loss_scratch_bytes(la.n_rays);
        if (sb != (size_t)(1 + (500 + 255) / 256) * 8 || adfp_mapper_loss_scratch_bytes(-3) != 0) { printf("FAIL adfp_mapper_loss_scratch_bytes\n"); ++g_fail; }
        EXPECT_NEG(adfp_mapper_loss_step(&la, nullptr, sb, dev<int>(9), dev<float>(10), 5, lrs, 0.9f, 0.999f, nullptr, st));
        EXPECT_NEG(adfp_mapper_loss_step(&la, (char*)dev<double>(11) + 4, sb, dev<int>(9), dev<float>(10), 5, lrs, 0.9f, 0.999f, nullptr, st));
        EXPECT_CODE(adfp_mapper_loss_step(&la, dev<double>(11), sb - 8, dev<int>(9), dev<float>(10), 5, lrs, 0.9f, 0.999f, nullptr, st), ADFP_E_WORKSPACE);
        EXPECT_NEG(adfp_mapper_loss_step(&la, dev<double>(11), sb, dev<int>(9), dev<float>(10), 9, lrs, 0.9f, 0.999f, nullptr, st));
        EXPECT_NEG(adfp_mapper_loss_step(&la, dev<double>(11), sb, nullptr, dev<float>(10), 5, lrs, 0.9f, 0.999f, nullptr, st));
        EXPECT_NEG(adfp_mapper_loss_step(nullptr, dev<double>(11), sb, dev<int>(9), dev<float>(10), 5, lrs, 0.9f, 0.999f, nullptr, st));
        EXPECT_REACHES_LAUNCH(adfp_mapper_loss_step(&la, dev<double>(11), sb, dev<int>(9), dev<float>(10), 5, lrs, 0.9f, 0.999f, dev<int>(12), st));
        EXPECT_REACHES_LAUNCH(adfp_mapper_loss_step(&la, dev<double>(11), sb, nullptr, nullptr, 0, nullptr, 0.9f, 0.999f, nullptr, st));
    }
    adfp_track_loss_args tl; memset(&tl, 0, sizeof(tl));
    tl.n_rays = 200; tl.handle_dynamic = 1; tl.w_color_loss = 0.5f; tl.depth = dev<double>(1); tl.uncertainty = dev<double>(2); tl.color = dev<float>(3); tl.gt_depth = dev<float>(4);
    tl.gt_color = dev<float>(5); tl.loss = dev<double>(6); tl.g_depth = dev<double>(7); tl.g_color = dev<float>(8);
    EXPECT_NEG(adfp_tracker_loss(nullptr, st));
    { adfp_track_loss_args t2 = tl; t2.n_rays = 100000; EXPECT_NEG(adfp_tracker_loss(&t2, st)); t2 = tl; t2.uncertainty = nullptr; EXPECT_NEG(adfp_tracker_loss(&t2, st)); t2 = tl; t2.loss = nullptr; EXPECT_NEG(adfp_tracker_loss(&t2, st)); }
    EXPECT_REACHES_LAUNCH(adfp_tracker_loss(&tl, st));
    EXPECT_NEG(adfp_camera_from_tensor(nullptr, dev<float>(1), st)); EXPECT_REACHES_LAUNCH(adfp_camera_from_tensor(dev<float>(2), dev<float>(1), st));
    EXPECT_NEG(adfp_camera_from_tensor_backward(dev<float>(2), nullptr, dev<float>(1), st)); EXPECT_REACHES_LAUNCH(adfp_camera_from_tensor_backward(dev<float>(2), dev<float>(3), dev<float>(1), st));
    EXPECT_NEG(adfp_select_pixels(dev<long long>(1), 100, 5, 4, 0, 64, 48, 64, dev<float>(2), dev<float>(3), dev<float>(4), dev<float>(5), dev<float>(6), dev<float>(7), st));      // H1 <= H0
    EXPECT_NEG(adfp_select_pixels(dev<long long>(1), 100, 0, 49, 0, 64, 48, 64, dev<float>(2), dev<float>(3), dev<float>(4), dev<float>(5), dev<float>(6), dev<float>(7), st));     // window beyond the image
    EXPECT_REACHES_LAUNCH(adfp_select_pixels(dev<long long>(1), 100, 4, 44, 4, 60, 48, 64, dev<float>(2), dev<float>(3), dev<float>(4), dev<float>(5), dev<float>(6), dev<float>(7), st));
    {   // the Tracker iteration's head and tail
        adfp_tracker_head_args h; memset(&h, 0, sizeof(h));
        h.cam = dev<float>(1); h.c2w = dev<float>(2); h.idx = dev<long long>(3); h.n = 100; h.H0 = 4; h.H1 = 44; h.W0 = 4; h.W1 = 60; h.H = 48; h.W = 64;
        h.depth_img = dev<float>(4); h.color_img = dev<float>(5); h.fx = h.fy = 50.f; h.cx = 32.f; h.cy = 24.f; h.bound = dev<double>(6);
        h.pix_i = dev<float>(7); h.pix_j = dev<float>(8); h.gt_depth = dev<float>(9); h.gt_color = dev<float>(10); h.rays_o = dev<float>(11); h.rays_d = dev<float>(12);
        h.keep = dev<unsigned char>(13); h.depth_max = dev<float>(14);
        EXPECT_NEG(adfp_tracker_head(nullptr, st));
        { adfp_tracker_head_args b = h; b.H1 = 49; EXPECT_NEG(adfp_tracker_head(&b, st)); }
        { adfp_tracker_head_args b = h; b.cam = nullptr; EXPECT_NEG(adfp_tracker_head(&b, st)); }
        { adfp_tracker_head_args b = h; b.rays_d = nullptr; EXPECT_NEG(adfp_tracker_head(&b, st)); }
        { adfp_tracker_head_args b = h; b.n = -1; EXPECT_NEG(adfp_tracker_head(&b, st)); }
        EXPECT_REACHES_LAUNCH(adfp_tracker_head(&h, st));
        adfp_tracker_tail_args t; memset(&t, 0, sizeof(t));
        t.pix_i = dev<float>(7); t.pix_j = dev<float>(8); t.n = 100; t.fx = t.fy = 50.f; t.cx = 32.f; t.cy = 24.f; t.g_rays_o = dev<float>(15); t.g_rays_d = dev<float>(16);
        t.cam = dev<float>(1); t.g_c2w = dev<float>(17); t.g_cam = dev<float>(18);
        EXPECT_NEG(adfp_tracker_tail(nullptr, st));
        { adfp_tracker_tail_args b = t; b.g_cam = nullptr; EXPECT_NEG(adfp_tracker_tail(&b, st)); }
        EXPECT_REACHES_LAUNCH(adfp_tracker_tail(&t, st));                     // gradients only
        t.step = 1;
        EXPECT_NEG(adfp_tracker_tail(&t, st));                                // a step without optimiser state
        t.exp_avg = dev<float>(19); t.exp_avg_sq = dev<float>(20); t.steps = dev<int>(21); t.derived = dev<float>(22); t.n_groups = 2; t.lr[0] = 1e-3f; t.lr[1] = 2e-4f;
        t.beta1 = 0.9f; t.beta2 = 0.999f; t.eps = 1e-8f; t.loss = dev<double>(23); t.best_loss = dev<double>(24); t.best_cam = dev<float>(25);
        { adfp_tracker_tail_args b = t; b.n_groups = 3; EXPECT_NEG(adfp_tracker_tail(&b, st)); }
        EXPECT_REACHES_LAUNCH(adfp_tracker_tail(&t, st));
    }
    {
        adfp_keyframe kf[2];
        memset(kf, 0, sizeof(kf));
        for (int f = 0; f < 2; ++f) { kf[f].idx = dev<long long>(1 + f); kf[f].depth_img = dev<float>(3); kf[f].color_img = dev<float>(4); }
        kf[1].c2w = dev<float>(5);
        EXPECT_NEG(adfp_sample_keyframes(2, kf, 100, 5, 4, 0, 64, 48, 64, 50.f, 50.f, 32.f, 24.f, dev<float>(6), dev<float>(7), dev<float>(8), dev<float>(9), st));       // H1 <= H0
        EXPECT_NEG(adfp_sample_keyframes(ADFP_KEYFRAMES_MAX + 1, kf, 100, 0, 48, 0, 64, 48, 64, 50.f, 50.f, 32.f, 24.f, dev<float>(6), dev<float>(7), dev<float>(8), dev<float>(9), st));
        EXPECT_NEG(adfp_sample_keyframes(2, nullptr, 100, 0, 48, 0, 64, 48, 64, 50.f, 50.f, 32.f, 24.f, dev<float>(6), dev<float>(7), dev<float>(8), dev<float>(9), st));
        EXPECT_NEG(adfp_sample_keyframes(2, kf, 100, 0, 48, 0, 64, 48, 64, 50.f, 50.f, 32.f, 24.f, nullptr, dev<float>(7), dev<float>(8), dev<float>(9), st));
        kf[1].idx = nullptr;
        EXPECT_NEG(adfp_sample_keyframes(2, kf, 100, 0, 48, 0, 64, 48, 64, 50.f, 50.f, 32.f, 24.f, dev<float>(6), dev<float>(7), dev<float>(8), dev<float>(9), st));
        kf[1].idx = dev<long long>(2);
        EXPECT_CODE(adfp_sample_keyframes(0, kf, 100, 0, 48, 0, 64, 48, 64, 50.f, 50.f, 32.f, 24.f, dev<float>(6), dev<float>(7), dev<float>(8), dev<float>(9), st), 0);
        EXPECT_REACHES_LAUNCH(adfp_sample_keyframes(2, kf, 100, 4, 44, 4, 60, 48, 64, 50.f, 50.f, 32.f, 24.f, dev<float>(6), dev<float>(7), dev<float>(8), dev<float>(9), st));
    }
    EXPECT_NEG(adfp_track_keep_best(nullptr, dev<float>(1), dev<double>(2), dev<float>(3), st));
    EXPECT_REACHES_LAUNCH(adfp_track_keep_best(dev<double>(4), dev<float>(1), dev<double>(2), dev<float>(3), st));
    EXPECT_NEG(adfp_sort_pairs(dev<int>(1), dev<int>(2), dev<int>(3), dev<int>(4), 1000, 0, dev<void>(5), 1 << 20, st));
    EXPECT_NEG(adfp_sort_pairs(dev<int>(1), dev<int>(2), dev<int>(3), dev<int>(4), 1000, 32, dev<void>(5), 1 << 20, st));
    EXPECT_CODE(adfp_sort_pairs(dev<int>(1), dev<int>(2), dev<int>(3), dev<int>(4), 1000, 20, dev<void>(5), 16, st), ADFP_E_WORKSPACE);
    EXPECT_CODE(adfp_sort_pairs(dev<int>(1), dev<int>(2), dev<int>(3), dev<int>(4), 0x7fffffffll, 20, dev<void>(5), (size_t)1 << 40, st), ADFP_E_UNSUPPORTED);
    EXPECT_REACHES_LAUNCH(adfp_sort_pairs(dev<int>(1), dev<int>(2), dev<int>(3), dev<int>(4), 100000, 24, dev<void>(5), adfp_sort_workspace_bytes(100000), st));
    const float origin[3] = {0, 0, 0}, intr[9] = {60, 0, 32, 0, 60, 24, 0, 0, 1};
    EXPECT_NEG(adfp_tsdf_integrate(nullptr, dev<float>(1), dev<float>(2), 64, 64, 64, origin, 0.02f, intr, c2w, dev<float>(3), dev<float>(4), 48, 64, 0.1f, 1.f, st));
    EXPECT_NEG(adfp_tsdf_integrate(dev<float>(5), dev<float>(1), dev<float>(2), 64, 64, 0, origin, 0.02f, intr, c2w, dev<float>(3), dev<float>(4), 48, 64, 0.1f, 1.f, st));
    EXPECT_CODE(adfp_tsdf_integrate(dev<float>(5), dev<float>(1), dev<float>(2), 2048, 2048, 2048, origin, 0.02f, intr, c2w, dev<float>(3), dev<float>(4), 48, 64, 0.1f, 1.f, st), ADFP_E_UNSUPPORTED);
    EXPECT_REACHES_LAUNCH(adfp_tsdf_integrate(dev<float>(5), dev<float>(1), nullptr, 61, 64, 63, origin, 0.02f, intr, c2w, nullptr, dev<float>(4), 48, 64, 0.1f, 1.f, st));   // ragged dims, no colour
    EXPECT_REACHES_LAUNCH(adfp_tsdf_integrate(dev<float>(5), dev<float>(1), dev<float>(2), 64, 64, 64, origin, 0.02f, intr, c2w, dev<float>(3), dev<float>(4), 48, 64, 0.1f, 1.f, st));

    EXPECT_NEG(adfp_ray_sort_keys(dev<float>(1), nullptr, dev<float>(2), 100, bound, dev<int>(3), dev<int>(4), st));
    { const double flat[3][2] = {{0, 1}, {2, 2}, {0, 1}}; EXPECT_NEG(adfp_ray_sort_keys(dev<float>(1), dev<float>(5), dev<float>(2), 100, flat, dev<int>(3), dev<int>(4), st)); }
    EXPECT_REACHES_LAUNCH(adfp_ray_sort_keys(dev<float>(1), dev<float>(5), nullptr, 131072, bound, dev<int>(3), dev<int>(4), st));

    EXPECT_NEG(adfp_ray_order_probe(dev<float>(1), dev<float>(5), nullptr, 1000, 0.f, dev<int>(3), st));
    EXPECT_NEG(adfp_ray_order_probe(dev<float>(1), dev<float>(5), nullptr, 1000, 0.1f, nullptr, st));
    EXPECT_REACHES_LAUNCH(adfp_ray_order_probe(dev<float>(1), dev<float>(5), dev<float>(2), 131072, 0.1f, dev<int>(3), st));

    // ---- the sharded render's gather
    const void* src[3] = {dev<void>(1), dev<void>(2), dev<void>(3)}; void* dst[3] = {dev<void>(4), dev<void>(5), dev<void>(6)};
    const int words[3] = {2, 2, 3}; const long long per[8] = {35650, 35650, 35650, 35650, 35650, 35650, 35650, 35649};
    EXPECT_NEG(adfp_gather_pack(0, src, words, 100, dev<void>(7), st)); EXPECT_NEG(adfp_gather_pack(9, src, words, 100, dev<void>(7), st));
    EXPECT_NEG(adfp_gather_pack(3, src, words, -1, dev<void>(7), st)); EXPECT_NEG(adfp_gather_pack(3, src, words, 100, nullptr, st));
    EXPECT_CODE(adfp_gather_pack(3, src, words, 0, dev<void>(7), st), 0);
    EXPECT_REACHES_LAUNCH(adfp_gather_pack(3, src, words, 35650, dev<void>(7), st));
    EXPECT_NEG(adfp_gather_unpack(3, dst, words, 0, 35650, per, dev<void>(7), st)); EXPECT_NEG(adfp_gather_unpack(3, dst, words, 65, 35650, per, dev<void>(7), st));
    EXPECT_NEG(adfp_gather_unpack(3, dst, words, 8, 35649, per, dev<void>(7), st));          // a rank with more rows than the padding
    EXPECT_REACHES_LAUNCH(adfp_gather_unpack(3, dst, words, 8, 35650, per, dev<void>(7), st));

    printf("host sanitizer driver: %d failed expectations\n", g_fail);
    return g_fail ? 1 : 0;
}
