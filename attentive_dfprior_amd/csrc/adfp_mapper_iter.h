// adfp_mapper_iter.h -- the Mapper's per-iteration glue on the device (reference src/Mapper.py:438-473), so that one
// optimisation iteration is a fixed sequence of kernels with no host read-back and can be captured into a HIP graph:
//
//   k_prefilter_mask   the bounding-box pre-filter as a per-ray keep flag + the max sensor depth of the KEPT rays
//                      (src/Mapper.py:438-449 compacts the batch with boolean indexing, which synchronises and makes the
//                      batch size data dependent; a dropped ray that is rendered anyway and masked out of the loss gives
//                      the kept rays the same outputs and the same gradients, because `far` only sees the batch
//                      through max(gt_depth), Renderer.py:159, :195)
//   k_mapper_loss      the three L1 terms of src/Mapper.py:457-469 and their cotangents (torch's abs backward = sign)
//   k_adam_prep        step += 1 and the bias corrections of torch.optim.Adam for every parameter group, on the device
//   k_masked_adam_dev  k_masked_adam with the step-dependent scalars read from device memory
#pragma once
#include "adfp_device.h"

__global__ __launch_bounds__(1024) void k_prefilter_mask(const float* __restrict__ ro, const float* __restrict__ rd,
                                                         const float* __restrict__ depth, int n, const double* __restrict__ bnd,
                                                         unsigned char* __restrict__ keep, float* __restrict__ depth_max) {
    __shared__ float s_m[16];
    double b[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) b[k] = bnd[k];
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 1024) {
        double t = INFINITY; bool nan = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double o = (double)ro[3 * i + k], d = (double)rd[3 * i + k];
            const double t0 = (b[2 * k] - o) / d, t1 = (b[2 * k + 1] - o) / d;
            nan |= (t0 != t0) | (t1 != t1);            // torch.max / torch.min propagate NaN
            const double tm = t0 > t1 ? t0 : t1;
            t = tm < t ? tm : t;
        }
        const float dep = depth[i];
        const bool k_ = !nan && (t >= (double)dep);
        keep[i] = k_ ? 1 : 0;
        if (k_) mx = dep > mx ? dep : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float v = __shfl_xor(mx, o); mx = v > mx ? v : mx; }
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = s_m[0];
        for (int w = 1; w < 16; ++w) m = s_m[w] > m ? s_m[w] : m;
        *depth_max = m;
    }
}

#define ADFP_ADAM_MAX_GROUPS 8
struct AdamPrepArgs { int* steps; float* derived; int n; float beta1, beta2; float lr[ADFP_ADAM_MAX_GROUPS]; const int* skip; };
struct LossArgs {
    int n, S, color_term, warmup;
    float w_color;
    const double* depth; const float* color; const float* weight;
    const float* gt_depth; const float* gt_color; const unsigned char* keep;
    double* loss; double* g_depth; float* g_color; float* g_weight;
    // adfp_mapper_loss_step: the loss is WRITTEN (per-workgroup partial sums in scratch + 8, added up in order by the workgroup that
    // draws the last ticket from the int at scratch, which it leaves zero again), and workgroup 0 does k_adam_prep's work on the side
    double* scratch;
    AdamPrepArgs prep;             // prep.n == 0: none
};
ADFP_DEV void adam_prep_group(const AdamPrepArgs& a, int g) {
    if (a.skip && *a.skip) {          // the iteration's gradients are not valid (f16-range repair): nobody steps, see masked_adam_block
        a.derived[2 * g] = 0.f; a.derived[2 * g + 1] = 0.f;
        return;
    }
    if (a.lr[g] < 0.f) return;
    const int t = a.steps[g] + 1;
    a.steps[g] = t;
    const double bc1 = 1.0 - pow((double)a.beta1, (double)t), bc2 = 1.0 - pow((double)a.beta2, (double)t);
    a.derived[2 * g] = (float)((double)a.lr[g] / bc1);
    a.derived[2 * g + 1] = (float)sqrt(bc2);
}
ADFP_DEV float sign_f(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }     // torch.sign: 0 at 0, NaN -> 0 here
// One thread per ray for the depth and colour terms, then all threads stride over the N x S attention weights of the warm-up
// term; the loss is summed in registers and leaves a wave through ONE f64 atomic (one atomic per ray serialised 5 000 adders on
// one address: 63 us per call; a wave walking its rays one after the other was latency bound: 19 us).
__global__ __launch_bounds__(256) void k_mapper_loss(LossArgs a) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * 256 + threadIdx.x;
    double part = 0.0;
    if (ray < a.n) {
        const bool kept = !a.keep || a.keep[ray];
        const float gd = a.gt_depth[ray];
        double g = 0.0;
        if (kept && gd > 0.f) {                                        // depth_mask = batch_gt_depth > 0, Mapper.py:457
            const double diff = (double)gd - a.depth[ray];            // f32 - f64 -> f64
            part += diff < 0 ? -diff : diff;
            g = diff > 0 ? -1.0 : (diff < 0 ? 1.0 : 0.0);              // d|gt - d|/dd = -sign(gt - d)
        }
        a.g_depth[ray] = g;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float gk = 0.f;
            if (a.color_term && kept) {                                // Mapper.py:466-469
                const float diff = a.gt_color[3 * ray + k] - a.color[3 * ray + k];
                part += (double)(a.w_color * fabsf(diff));
                gk = -a.w_color * sign_f(diff);
            }
            if (a.g_color) a.g_color[3 * ray + k] = gk;
        }
    }
    if (a.g_weight) {
        const long long total = (long long)a.n * a.S, stride = (long long)gridDim.x * 256;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
            float g = 0.f;
            if (a.warmup) {                                            // + |weight - 1|.sum(), Mapper.py:459-461
                const int r = (int)(i / a.S);
                if (!a.keep || a.keep[r]) {
                    const float diff = a.weight[i] - 1.f;
                    part += (double)fabsf(diff);
                    g = sign_f(diff);
                }
            }
            a.g_weight[i] = g;
        }
    }
    if (a.scratch) {
        // one launch instead of three (a zero fill of the loss word, this kernel, k_adam_prep): ~5 us each inside a graph replay
        if (blockIdx.x == 0 && (int)threadIdx.x < a.prep.n) adam_prep_group(a.prep, (int)threadIdx.x);
        __shared__ double s_part[4];
        __shared__ int s_last;
        part = wave_sum(part);
        if (lane == 0) s_part[threadIdx.x >> 6] = part;
        __syncthreads();
        // Cross-workgroup hand-off without fences (MI355X_MICROARCH.md, "hand-offs measured with sc1 loads", first row: a
        // __threadfence() is ~3.5 us, most of this kernel): ONE lane per workgroup stores its partial sum write-through (agent-scope
        // atomic store = sc1), waits for it, then adds to the ticket (agent-scope atomic); the workgroup whose add came last -- told by
        // the returned value -- reads the partial sums with agent-scope loads (sc1: past its L1), its other lanes after the barrier.
        if (threadIdx.x == 0) {
            const double mine = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
            __hip_atomic_store((unsigned long long*)a.scratch + 1 + blockIdx.x, (unsigned long long)__double_as_longlong(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_last = __hip_atomic_fetch_add((int*)a.scratch, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
        }
        __syncthreads();
        if (s_last && threadIdx.x < 64) {          // the last workgroup: added up in workgroup order (reproducible)
            double t = 0.0;
            for (int b = lane; b < (int)gridDim.x; b += 64)
                t += __longlong_as_double((long long)__hip_atomic_load((unsigned long long*)a.scratch + 1 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            t = wave_sum(t);
            if (lane == 0) {
                if (a.loss) *a.loss = t;
                __hip_atomic_store((int*)a.scratch, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the ticket is zero again for the next call
            }
        }
    } else if (a.loss) {
        part = wave_sum(part);
        if (lane == 0 && part != 0.0) atomicAdd(a.loss, part);
    }
}

// Per parameter group g < n with lr[g] >= 0:  t = ++steps[g],  derived[g] = { lr[g] / (1 - beta1^t), sqrt(1 - beta2^t) } -- the
// python-float (double) arithmetic of torch.optim.Adam, rounded to f32 once.  A negative lr marks a group that does not step
// in this iteration (torch skips parameters without a gradient).  One launch for all groups.
__global__ void k_adam_prep(AdamPrepArgs a) {
    const int g = threadIdx.x;
    if (blockIdx.x != 0 || g >= a.n) return;
    adam_prep_group(a, g);
}
