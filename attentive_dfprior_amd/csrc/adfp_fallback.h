// adfp_fallback.h -- the f32 repair path of a forward call whose f16-split kernels met an operand outside the f16 range.
//
// The f16-split decoders (adfp_decode_h.h) cannot represent |x| >= 65504; they detect it (every split value is range-checked,
// the weights at pack time) and raise the call's flag word.  k_fallback_points is launched after them in EVERY call that runs
// f16-split kernels and has the flat parameters at hand (adfp_scene.flat_*): it reads the flag and returns at once when it is
// clear -- the price of the common path is one empty launch -- and otherwise re-evaluates every point of the call in plain f32
// (one thread per point, weights read from the flat state_dict-order buffers through the scalar cache, fma chains in the
// reference's order) and overwrites raw / w / att_occ before the compositor or the caller reads them.  No host involvement, so
// the repair also works inside a captured HIP graph.  It is a slow path on purpose (tens of milliseconds for a 100 000-ray
// batch): the sticky status word tells the host which network tripped, and the host hands over that network's exact image from
// the next call on.  The training state (ReLU masks, layer inputs) of a repaired call is NOT rebuilt; the backward entries read
// the same flag and return zero gradients for it (adfp_train_state.counter[8]).
//
// Follows DF.forward (reference src/conv_onet/models/decoder.py:307-353) + the bound rule of Renderer.eval_points
// (src/utils/Renderer.py:51-64) exactly as the fast kernels do.
#pragma once
#include "adfp_device.h"

// all 32 channels of a channels-last grid at normalised position pn (the scalar twin of gather16)
ADFP_DEV void fb_gather32(const GridDev& g, const float pn[3], float* __restrict__ c) {
    int xi[2], yi[2], zi[2]; float wx[2], wy[2], wz[2];
    tri_axis(pn[0], g.X, xi[0], xi[1], wx[0], wx[1]);
    tri_axis(pn[1], g.Y, yi[0], yi[1], wy[0], wy[1]);
    tri_axis(pn[2], g.Z, zi[0], zi[1], wz[0], wz[1]);
    for (int k = 0; k < 32; ++k) c[k] = 0.f;
    for (int dz = 0; dz < 2; ++dz)
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                const float w = (wx[dx] * wy[dy]) * wz[dz];
                const float* src = g.data + (((long long)zi[dz] * g.Y + yi[dy]) * g.X + xi[dx]) * 32;
                for (int k = 0; k < 32; ++k) c[k] = fmaf(src[k], w, c[k]);
            }
}

// MLP.forward (decoder.py:177-203) from the flat parameters: e = sin(pf @ B); h = relu(W_i h + b_i) + (Wc_i c + bc_i); skip-concat
// [e, h] into layer 3; out = Wo h + bo
template <int CDIM, int NOUT>
ADFP_DEV void fb_decoder(const float* __restrict__ flat, const float* __restrict__ c, const float pf[3], float* __restrict__ out) {
    using F = DecLayout<CDIM, NOUT>;
    float e[93], h[32], hn[32];
    for (int j = 0; j < 93; ++j) {
        const float arg = fmaf(pf[2], flat[F::F_EB + 2 * 93 + j], fmaf(pf[1], flat[F::F_EB + 93 + j], pf[0] * flat[F::F_EB + j]));
        e[j] = adfp_sinf(arg);
    }
    for (int i = 0; i < 5; ++i) {
        const int ind = F::in_dim(i);
        const float* W = flat + F::F_PL(i);
        const float* Wc = flat + F::F_FC(i);
        for (int u = 0; u < 32; ++u) {
            float s = W[32 * ind + u];                                  // bias
            if (i == 0 || i == 3) for (int j = 0; j < 93; ++j) s = fmaf(W[u * ind + j], e[j], s);
            if (i == 3) for (int j = 0; j < 32; ++j) s = fmaf(W[u * ind + 93 + j], h[j], s);
            if (i == 1 || i == 2 || i == 4) for (int j = 0; j < 32; ++j) s = fmaf(W[u * ind + j], h[j], s);
            float t = Wc[32 * CDIM + u];
            for (int k = 0; k < CDIM; ++k) t = fmaf(Wc[u * CDIM + k], c[k], t);
            hn[u] = fmaxf(s, 0.f) + t;
        }
        for (int u = 0; u < 32; ++u) h[u] = hn[u];
    }
    for (int o = 0; o < NOUT; ++o) {
        float s = flat[F::F_OB + o];
        for (int j = 0; j < 32; ++j) s = fmaf(flat[F::F_OW + o * 32 + j], h[j], s);
        out[o] = s;
    }
}

// mlp_tsdf.forward (decoder.py:240-258) on (occ, u = inv_tsdf): fused occupancy and the attention weight a1
ADFP_DEV void fb_attention(const float* __restrict__ flat, float occ, float u, float& fused, float& a1) {
    using A = AttLayout;
    float x[128], y[128];
    for (int k = 0; k < 64; ++k) x[k] = fmaxf(fmaf(u, flat[A::F_W0 + 2 * k + 1], fmaf(occ, flat[A::F_W0 + 2 * k], flat[A::F_B0 + k])), 0.f);
    for (int r = 0; r < 128; ++r) { float s = flat[A::F_B1 + r]; for (int k = 0; k < 64; ++k) s = fmaf(flat[A::F_W1 + r * 64 + k], x[k], s); y[r] = fmaxf(s, 0.f); }
    for (int r = 0; r < 128; ++r) { float s = flat[A::F_B2 + r]; for (int k = 0; k < 128; ++k) s = fmaf(flat[A::F_W2 + r * 128 + k], y[k], s); x[r] = fmaxf(s, 0.f); }
    for (int r = 0; r < 64; ++r) { float s = flat[A::F_B3 + r]; for (int k = 0; k < 128; ++k) s = fmaf(flat[A::F_W3 + r * 128 + k], x[k], s); y[r] = fmaxf(s, 0.f); }
    float l0 = flat[A::F_BO], l1 = flat[A::F_BO + 1];
    for (int k = 0; k < 64; ++k) { l0 = fmaf(flat[A::F_WO + k], y[k], l0); l1 = fmaf(flat[A::F_WO + 64 + k], y[k], l1); }
    const float m = fmaxf(l0, l1);
    const float e0 = expf(l0 - m), e1 = expf(l1 - m);
    const float den = e0 + e1;
    const float a0 = e0 / den;
    a1 = e1 / den;
    fused = a0 * occ + a1 * u;
}

struct FallbackArgs {
    PtsDev P; NormDev nb; double b[6];
    GridDev low, high, color;
    const float* flat_low; const float* flat_high; const float* flat_color; const float* flat_att;
    int stage, apply_bound;
    const unsigned char* flags;    // stage >= high: ADFP_F_INBOUND | ADFP_F_BAND per point (k_tsdf)
    const int* list; const int* count_ptr; const float* att_u;   // the in-band list and its inv_tsdf values (k_tsdf)
    float* att_occ;                // per list entry: high + low (the backward's state)
    float* raw; float* w;
    const int* call_flag;
};

// Thread t < P: point t (LOW, COLOR; final occupancy of points outside the band).  Thread P + i: in-band list entry i (LOW again,
// HIGH, attention; final occupancy and weight of that point).  The two kinds write disjoint words.
__global__ __launch_bounds__(256) void k_fallback_points(FallbackArgs a) {
    if (*a.call_flag == 0) return;
    const bool fuse = a.stage != ADFP_STAGE_LOW;
    const long long total = (long long)a.P.n + (fuse ? a.P.n : 0);
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const bool entry = t >= a.P.n;
        int q = (int)t, idx = 0;
        if (entry) {
            idx = (int)(t - a.P.n);
            if (idx >= *a.count_ptr) continue;
            q = a.list[idx];
        }
        double pt[3]; float pn[3], pf[3], c[64];
        load_point(a.P, q, pt);
        normalize3(a.nb, pt, pn);
        pf[0] = (float)pt[0]; pf[1] = (float)pt[1]; pf[2] = (float)pt[2];
        const bool pnan = (pt[0] != pt[0]) | (pt[1] != pt[1]) | (pt[2] != pt[2]);
        const unsigned f = a.flags ? a.flags[q] : 0u;
        const bool inb = in_bound(pt, a.b);
        float low;
        fb_gather32(a.low, pn, c + 32);
        fb_decoder<32, 1>(a.flat_low, c + 32, pf, &low);
        if (pnan) low = __builtin_nanf("");
        if (!entry) {
            if (!fuse || !(f & ADFP_F_BAND)) {
                a.raw[4ll * q + 3] = (inb || !a.apply_bound) ? low : 100.f;                       // Renderer.py:64
                a.w[q] = 1.f;
            }
            if (a.stage == ADFP_STAGE_COLOR) {
                float rgb[4];
                fb_gather32(a.color, pn, c);
                fb_decoder<32, 4>(a.flat_color, c, pf, rgb);
                for (int k = 0; k < 3; ++k) a.raw[4ll * q + k] = pnan ? __builtin_nanf("") : rgb[k];
            }
        } else {
            float high, fused, a1;
            fb_gather32(a.high, pn, c);                                                          // [high features, low features], decoder.py:182-187
            fb_decoder<64, 1>(a.flat_high, c, pf, &high);
            if (pnan) high = __builtin_nanf("");
            const float occ = high + low;                                                        // decoder.py:342
            a.att_occ[idx] = occ;
            fb_attention(a.flat_att, occ, a.att_u[idx], fused, a1);
            a.raw[4ll * q + 3] = (inb || !a.apply_bound) ? fused : 100.f;
            a.w[q] = a1;
        }
    }
}
