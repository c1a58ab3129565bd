// adfp_tracker_iter.h -- the Tracker's per-iteration glue on the device (reference src/Tracker.py:75-134,
// Tracker.optimize_cam_in_batch), so that one camera-tracking iteration is a fixed sequence of kernels with no host read-back and
// can be replayed from a HIP graph (tracking.TrackerIteration), like adfp_mapper_iter.h does for the Mapper:
//
//   k_camera_from_tensor(_bwd)   quaternion + translation -> camera-to-world and back (src/common.py:139-178)
//   k_select_pixels              the sampled pixels' coordinates, sensor depth and colour (src/common.py:94-124)
//   k_tracker_loss               the tracking loss with its median-based outlier mask and the cotangents (src/Tracker.py:116-129)
//   k_keep_best                  the running "candidate_cam_tensor" of the iteration loop (src/Tracker.py:261-263)
//
// The rays (adfp_rays_from_uv), the pre-filter (adfp_prefilter_mask), the render forward / backward with ray gradients and the
// Adam step (adfp_adam_prep + adfp_masked_adam_multi on the 7 pose parameters) are the existing entries.
#pragma once
#include "adfp_device.h"

// R = I + two_s M(q), two_s = 2 / |q|^2, q = (r, i, j, k): quad2rotation of src/common.py:139-163, float32 like the reference
ADFP_DEV void quat_terms(const float* q, float& two_s, float M[9]) {
    const float qr = q[0], qi = q[1], qj = q[2], qk = q[3];
    two_s = 2.0f / (((qr * qr + qi * qi) + qj * qj) + qk * qk);
    M[0] = -(qj * qj + qk * qk); M[1] = qi * qj - qk * qr;   M[2] = qi * qk + qj * qr;
    M[3] = qi * qj + qk * qr;    M[4] = -(qi * qi + qk * qk); M[5] = qj * qk - qi * qr;
    M[6] = qi * qk - qj * qr;    M[7] = qj * qk + qi * qr;   M[8] = -(qi * qi + qj * qj);
}
ADFP_DEV void camera_from_tensor_dev(const float* __restrict__ cam, float* c2w /*[16]*/) {
    float two_s, M[9];
    quat_terms(cam, two_s, M);
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) c2w[4 * a + b] = (a == b ? 1.0f : 0.0f) + two_s * M[3 * a + b];
        c2w[4 * a + 3] = cam[4 + a];
    }
    c2w[12] = 0.f; c2w[13] = 0.f; c2w[14] = 0.f; c2w[15] = 1.f;
}
__global__ void k_camera_from_tensor(const float* __restrict__ cam, float* __restrict__ c2w) {
    if (blockIdx.x || threadIdx.x) return;
    camera_from_tensor_dev(cam, c2w);
}
// g_cam[0..3] = dL/dq, g_cam[4..6] = dL/dT from dL/d c2w (rows 0-2):
//   dL/dq_m = two_s sum_ab G_ab dM_ab/dq_m - two_s^2 q_m sum_ab G_ab M_ab       (d two_s / d q_m = -two_s^2 q_m)
ADFP_DEV void camera_from_tensor_bwd_dev(const float* __restrict__ cam, const float* g_c2w, float* g_cam);
__global__ void k_camera_from_tensor_bwd(const float* __restrict__ cam, const float* __restrict__ g_c2w, float* __restrict__ g_cam) {
    if (blockIdx.x || threadIdx.x) return;
    camera_from_tensor_bwd_dev(cam, g_c2w, g_cam);
}
ADFP_DEV void camera_from_tensor_bwd_dev(const float* __restrict__ cam, const float* g_c2w, float* g_cam) {
    float two_s, M[9], G[9];
    quat_terms(cam, two_s, M);
    float gm = 0.f;
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { G[3 * a + b] = g_c2w[4 * a + b]; gm += G[3 * a + b] * M[3 * a + b]; }
    const float qr = cam[0], qi = cam[1], qj = cam[2], qk = cam[3];
    const float dr = -qk * G[1] + qj * G[2] + qk * G[3] - qi * G[5] - qj * G[6] + qi * G[7];
    const float di = qj * G[1] + qk * G[2] + qj * G[3] - 2.f * qi * G[4] - qr * G[5] + qk * G[6] + qr * G[7] - 2.f * qi * G[8];
    const float dj = -2.f * qj * G[0] + qi * G[1] + qr * G[2] + qi * G[3] + qk * G[5] - qr * G[6] + qk * G[7] - 2.f * qj * G[8];
    const float dk = -2.f * qk * G[0] - qr * G[1] + qi * G[2] + qr * G[3] - 2.f * qk * G[4] + qj * G[5] + qi * G[6] + qj * G[7];
    const float t2 = two_s * two_s * gm;
    g_cam[0] = two_s * dr - t2 * qr; g_cam[1] = two_s * di - t2 * qi; g_cam[2] = two_s * dj - t2 * qj; g_cam[3] = two_s * dk - t2 * qk;
    g_cam[4] = g_c2w[3]; g_cam[5] = g_c2w[7]; g_cam[6] = g_c2w[11];
}

// pixel k of the window [H0, H0 + Hw) x [W0, W0 + Ww) in row-major order (get_sample_uv's meshgrid flattened, src/common.py:112-124)
__global__ __launch_bounds__(256) void k_select_pixels(const long long* __restrict__ idx, int n, int H0, int W0, int Ww, int W,
                                                       const float* __restrict__ depth, const float* __restrict__ color,
                                                       float* __restrict__ pi, float* __restrict__ pj, float* __restrict__ gd, float* __restrict__ gc) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const long long k = idx[t];
    const int row = H0 + (int)(k / Ww), col = W0 + (int)(k % Ww);
    pi[t] = (float)col; pj[t] = (float)row;
    const long long p = (long long)row * W + col;
    gd[t] = depth[p];
    gc[3 * t] = color[3 * p]; gc[3 * t + 1] = color[3 * p + 1]; gc[3 * t + 2] = color[3 * p + 2];
}

// k_select_pixels + k_rays_from_uv for every keyframe of the Mapper's window in one launch (adfp_sample_keyframes): thread t = ray
// t % n of frame t / n.  The ray arithmetic is k_rays_from_uv's, operation by operation.
struct KeyframeJobs {
    const long long* idx[ADFP_KEYFRAMES_MAX]; const float* c2w[ADFP_KEYFRAMES_MAX]; const float* depth[ADFP_KEYFRAMES_MAX];
    const float* color[ADFP_KEYFRAMES_MAX]; float pose[ADFP_KEYFRAMES_MAX][12];
    int n_frames, n, H0, W0, Ww, W; float fx, fy, cx, cy;
    float* ro; float* rd; float* gd; float* gc;
};
__global__ __launch_bounds__(256) void k_sample_keyframes(KeyframeJobs a) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= a.n_frames * a.n) return;
    const int f = t / a.n;
    const long long k = a.idx[f][t - f * a.n];
    const int row = a.H0 + (int)(k / a.Ww), col = a.W0 + (int)(k % a.Ww);
    const long long p = (long long)row * a.W + col;
    a.gd[t] = a.depth[f][p];
    a.gc[3 * t] = a.color[f][3 * p]; a.gc[3 * t + 1] = a.color[f][3 * p + 1]; a.gc[3 * t + 2] = a.color[f][3 * p + 2];
    const float pi = (float)col, pj = (float)row;
    const float dx = (pi - a.cx) / a.fx, dy = -(pj - a.cy) / a.fy, dz = -1.f;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        float c0, c1, c2, c3;
        if (a.c2w[f]) { c0 = a.c2w[f][4 * m]; c1 = a.c2w[f][4 * m + 1]; c2 = a.c2w[f][4 * m + 2]; c3 = a.c2w[f][4 * m + 3]; }
        else { c0 = a.pose[f][4 * m]; c1 = a.pose[f][4 * m + 1]; c2 = a.pose[f][4 * m + 2]; c3 = a.pose[f][4 * m + 3]; }
        a.rd[3 * t + m] = __fadd_rn(__fadd_rn(__fmul_rn(dx, c0), __fmul_rn(dy, c1)), __fmul_rn(dz, c2));
        a.ro[3 * t + m] = c3;
    }
}

// loss = sum_mask |gt_d - d| / sqrt(unc + 1e-10)  +  w_color sum_mask |gt_c - c|,   mask = kept & (gt_d > 0) [& tmp < 10 median(tmp)]
// (handle_dynamic; torch.median = the lower middle element of the kept rays' tmp).  ONE workgroup: tracking batches are a few
// hundred to a few thousand rays.  uncertainty is detached in the reference (:115), so it gets no cotangent.
//
// The median is a radix SELECT over the values' bit patterns (tmp >= 0, so the IEEE-754 bits order like the values): ten bits
// per pass from the top, a 1 024-bin LDS histogram of the candidates that still share the selected prefix, one workgroup scan to
// find the bin holding the wanted rank.  The value is known as soon as one candidate is left (typically after three passes
// for ~1 000 distinct values: 9 exponent bits, then 2 + 8 mantissa bits, ...) or all 64 bits are fixed (ties: equal values, and
// only the VALUE matters).  A rank count over all pairs -- the first version -- cost 47 us at 1 000 rays, VALU-bound on f64
// compares; this is ~5 us.
#define ADFP_TRACK_MAX_RAYS 8192
// k_tracker_loss keeps the kept rays' 64-bit keys in LDS: 8192 x 8 B = 64 KB + a 4 KB histogram + scalars ~ 69 KB of static LDS.
// That is a gfx950 budget (160 KB per workgroup); gfx90a / gfx942 stop at 64 KB -- this library targets gfx950 only (build.sh).
static_assert(ADFP_TRACK_MAX_RAYS * 8 + 1024 * 4 + 16 * 4 + 16 * 8 + 64 <= 160 * 1024, "k_tracker_loss: static LDS beyond the gfx950 workgroup limit");
struct TrackLossArgs {
    int n, handle_dynamic; float w_color;
    const double* depth; const double* unc; const float* color; const float* gd; const float* gc; const unsigned char* keep;
    double* loss; double* g_depth; float* g_color;
};
__global__ __launch_bounds__(1024) void k_tracker_loss(TrackLossArgs a) {
    __shared__ unsigned long long s_key[ADFP_TRACK_MAX_RAYS];
    __shared__ int s_hist[1024];
    __shared__ int s_wtot[16];
    __shared__ double s_part[16];
    __shared__ unsigned long long s_medkey;
    __shared__ int s_kept, s_nan, s_digit, s_want, s_left;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_kept = 0; s_nan = 0; s_medkey = 0ull; }
    __syncthreads();
    int kept = 0, nans = 0;
    for (int i = threadIdx.x; i < a.n; i += 1024) {
        unsigned long long key = ~0ull;                        // a dropped ray sorts last and is never selected
        if (!a.keep || a.keep[i]) {
            const double diff = (double)a.gd[i] - a.depth[i];
            const double t = (diff < 0 ? -diff : diff) / sqrt(a.unc[i] + 1e-10);
            ++kept;
            if (t != t) ++nans;
            key = (unsigned long long)__double_as_longlong(t) & 0x7fffffffffffffffull;      // -0.0 -> +0.0
        }
        s_key[i] = key;
    }
    if (kept) atomicAdd(&s_kept, kept);
    if (nans) atomicAdd(&s_nan, nans);
    __syncthreads();
    const int K = s_kept;
    const bool poisoned = s_nan != 0;                          // torch.median propagates NaN: every comparison with it is false
    if (a.handle_dynamic && K > 0 && !poisoned) {
        unsigned long long prefix = 0ull, fixed = 0ull;        // the selected bits so far and their mask
        int want = (K - 1) >> 1, left = K;
        for (int shift = 54; ; shift -= 10) {
            const int bits = shift >= 0 ? 10 : 10 + shift;     // the last pass takes the remaining 4 bits
            const int sh = shift >= 0 ? shift : 0;
            s_hist[threadIdx.x] = 0;
            __syncthreads();
            for (int i = threadIdx.x; i < a.n; i += 1024) {
                const unsigned long long k = s_key[i];
                if (((k ^ prefix) & fixed) == 0ull && k != ~0ull) atomicAdd(&s_hist[(int)((k >> sh) & ((1u << bits) - 1u))], 1);
            }
            __syncthreads();
            // inclusive scan of the 1 024 bins, one per thread
            const int mine = s_hist[threadIdx.x];
            int inc = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
            if (lane == 63) s_wtot[wv] = inc;
            __syncthreads();
            int base = 0;
            for (int w = 0; w < wv; ++w) base += s_wtot[w];
            inc += base;
            if (mine > 0 && want >= inc - mine && want < inc) { s_digit = threadIdx.x; s_want = want - (inc - mine); s_left = mine; }
            __syncthreads();
            prefix |= (unsigned long long)s_digit << sh;
            fixed |= (unsigned long long)((1u << bits) - 1u) << sh;
            want = s_want; left = s_left;
            if (left == 1 || sh == 0) break;
        }
        // the candidates that are left all carry the median's value (one candidate, or equal values): any of them writes it
        for (int i = threadIdx.x; i < a.n; i += 1024) {
            const unsigned long long k = s_key[i];
            if (((k ^ prefix) & fixed) == 0ull && k != ~0ull) s_medkey = k;
        }
        __syncthreads();
    }
    const double med = poisoned ? (double)NAN : __longlong_as_double((long long)s_medkey);
    const double lim = 10.0 * med;
    double part = 0.0;
    for (int i = threadIdx.x; i < a.n; i += 1024) {
        const bool k_ = !a.keep || a.keep[i];
        double g = 0.0;
        float gcol[3] = {0.f, 0.f, 0.f};
        if (k_ && a.gd[i] > 0.f) {
            const double diff = (double)a.gd[i] - a.depth[i];
            const double rs = sqrt(a.unc[i] + 1e-10);
            const double t = (diff < 0 ? -diff : diff) / rs;
            if (!a.handle_dynamic || t < lim) {
                part += t;
                g = (diff > 0 ? -1.0 : (diff < 0 ? 1.0 : 0.0)) / rs;
                for (int c = 0; c < 3; ++c) {
                    const float dc = a.gc[3 * i + c] - a.color[3 * i + c];
                    part += (double)(a.w_color * fabsf(dc));
                    gcol[c] = -a.w_color * (dc > 0.f ? 1.f : (dc < 0.f ? -1.f : 0.f));
                }
            }
        }
        a.g_depth[i] = g;
        a.g_color[3 * i] = gcol[0]; a.g_color[3 * i + 1] = gcol[1]; a.g_color[3 * i + 2] = gcol[2];
    }
    part = wave_sum(part);
    if (lane == 0) s_part[wv] = part;
    __syncthreads();
    if (threadIdx.x == 0 && a.loss) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += s_part[w];
        *a.loss = s;
    }
}

// if loss < best_loss: best_loss = loss, best_cam = cam   (NaN compares false, like the reference's `if loss < current_min_loss`)
__global__ void k_keep_best(const double* __restrict__ loss, const float* __restrict__ cam, double* __restrict__ best_loss, float* __restrict__ best_cam) {
    if (blockIdx.x || threadIdx.x >= 7) return;
    const bool better = *loss < *best_loss;
    const float v = cam[threadIdx.x];
    __syncthreads();                               // every lane has read best_loss before lane 0 overwrites it
    if (!better) return;
    best_cam[threadIdx.x] = v;
    if (threadIdx.x == 0) *best_loss = *loss;
}
