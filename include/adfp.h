/*
 * adfp.h -- C ABI of libadfp.so: the MI355X (gfx950) implementation of the per-ray
 * volume-rendering hot path of MachinePerceptionLab/Attentive_DFPrior.
 *
 * The reference offers no C ABI, plugin or operator registry for this path; its boundary
 * is two Python objects (src/DF_Prior.py:50-51 `shared_decoders`, :111 `renderer`).  The
 * entry points below are what a ctypes binding under those two objects calls; each cites the
 * reference function it replaces.  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer owned by the caller (PyTorch) unless it says "host".
 *    The library never allocates, frees or retains device memory.
 *  - All work is enqueued asynchronously on `stream` (a hipStream_t passed as void*).
 *  - Return value: 0 = ok; <0 = argument error detected on the host (ADFP_E_*);
 *    >0 = hipError_t of a failed launch.  No exception crosses the ABI.
 *  - Feature grids are consumed channels-last ([Z][Y][X][32] fp32; one voxel = one 128-B
 *    line); adfp_relayout_grid converts from the reference's [1,32,Z,Y,X] layout
 *    (src/DF_Prior.py:243-264).  The TSDF is consumed in place through element strides, so the
 *    permuted view of get_tsdf.py:95-97 needs no copy.
 *  - Decoder weights are consumed as "packed images" (MFMA operand order)
 *    produced by adfp_pack_decoder / adfp_pack_attention from a flat fp32 buffer that is the
 *    concatenation of the module's parameters in state_dict order (decoder.py:110-166,
 *    :212-228).
 */
#ifndef ADFP_H
#define ADFP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADFP_VERSION 131

/* error codes (host-detected) */
#define ADFP_E_ARG        (-1)   /* null pointer / negative size */
#define ADFP_E_UNSUPPORTED (-2)  /* N_importance>0, occupancy=False, S too large ... */
#define ADFP_E_WORKSPACE  (-3)   /* workspace too small */

/* stage enum: DF.forward(stage=...) decoder.py:307 */
#define ADFP_STAGE_LOW   0
#define ADFP_STAGE_HIGH  1
#define ADFP_STAGE_COLOR 2

/* decoder kinds for packing */
#define ADFP_DEC_LOW   0   /* MLP(name='low',  c_dim=32, color=False) decoder.py:276 */
#define ADFP_DEC_HIGH  1   /* MLP(name='high', c_dim=64, concat_feature) decoder.py:279 */
#define ADFP_DEC_COLOR 2   /* MLP(name='color',c_dim=32, color=True) decoder.py:282 */

/* point source modes for adfp_eval_points */
#define ADFP_PTS_RAYS 0    /* p = rays_o[r] + rays_d[r] * z_vals[r][s]  (Renderer.py:223) */
#define ADFP_PTS_F64  1    /* explicit [P,3] float64 */
#define ADFP_PTS_F32  2    /* explicit [P,3] float32 */

#define ADFP_MAX_SAMPLES 256

typedef struct adfp_grid {
    const float* data;      /* channels-last [Z][Y][X][32] */
    int Z, Y, X;
} adfp_grid;

typedef struct adfp_tsdf {
    const float* data;      /* element (z,y,x) at data[z*sZ + y*sY + x*sX] */
    int Z, Y, X;
    long long sZ, sY, sX;   /* element strides (the reference's view has sZ=1) */
    /* Optional: the CORNER-BLOCK copy of the same volume (adfp_relayout_tsdf), or NULL.  When given, the TSDF stage of the render
     * path (a10 inside adfp_render_forward / adfp_eval_points / adfp_tsdf_stage) reads it instead of `data`: one aligned 32-byte
     * piece per sample wherever the sample lies -- for batches whose neighbouring rays are NOT neighbouring pixels (every
     * 8-corner lookup of the plain volume then costs four 64-byte sectors).  Same values bit for bit.  Every other consumer
     * (adfp_sample_tsdf, the backward's TSDF gradient) reads `data`. */
    const float* corner_blocks;
} adfp_tsdf;

/* Everything DF.forward reads besides the points: decoder.py:307-353. */
typedef struct adfp_scene {
    double bound[3][2];       /* Renderer.bound / MLP.bound   (src/DF_Prior.py:177-194), f64 */
    double tsdf_bnds[3][2];   /* tsdf_bnds                    (src/DF_Prior.py:86-91),   f64 */
    adfp_grid low, high, color;
    adfp_tsdf tsdf;
    const float* w_low;       /* packed images (adfp_pack_decoder / adfp_pack_attention) */
    const float* w_high;
    const float* w_color;
    const float* w_att;
    /* optional "H" images (adfp_pack_decoder_h): when non-NULL the FORWARD decoders run their MLP on
     * f16 MFMA with a 3-product split of every f32 operand (fp32-grade accuracy);
     * NULL = exact f32-input MFMA from w_*. */
    const void* h_low;
    const void* h_high;
    const void* h_color;
    const void* h_att;        /* adfp_pack_attention_h */
    /* optional "T" images (adfp_pack_decoder_ht: the transposed weights as f16 hi/lo halves).  With them, and with the ReLU
     * masks the training forward left in adfp_train_state, the decoder BACKWARD runs on f16 MFMA with the same 3-product
     * split and recomputes nothing; NULL (or no masks, or a ray / point gradient requested) = the exact f32 backward from
     * w_*. */
    const void* ht_low;
    const void* ht_high;
    const void* ht_color;
    const void* ht_att;       /* adfp_pack_attention_ht: the attention network's backward on f16 MFMA */
    /* Flat (state_dict order) parameters of the four networks, optional.  With them, a forward call whose f16-split kernels
     * met an operand outside the f16 range (|x| >= 65504: a weight, a grid feature or a hidden activation) REPAIRS ITSELF: a
     * predicated fallback kernel re-evaluates the call's points in plain f32 from these buffers before anything consumes the
     * outputs (no host involvement, graph-capturable).  NULL = no repair; the call then only reports (status). */
    const float* flat_low;
    const float* flat_high;
    const float* flat_color;
    const float* flat_att;
    /* Sticky status word the kernels OR into (system-scope atomic): device memory or device-visible pinned host memory,
     * NULL = none.  ADFP_STATUS_F16_RANGE_<net>: the f16-split kernels of that network met an operand outside the f16 range
     * in some call since the caller last cleared the word.  With flat_* set that call's outputs were repaired (see above); the
     * caller should hand over the exact image (w_*, h_* = NULL) for that network from now on -- the repair is a slow path.
     * ADFP_STATUS_F16_RANGE_BWD: the same in a backward kernel (that call's gradients are not reliable). */
    int* status;
} adfp_scene;
#define ADFP_STATUS_F16_RANGE_LOW   1
#define ADFP_STATUS_F16_RANGE_HIGH  2
#define ADFP_STATUS_F16_RANGE_COLOR 4
#define ADFP_STATUS_F16_RANGE_ATT   8
#define ADFP_STATUS_F16_RANGE_BWD   16
#define ADFP_STATUS_F16_RANGE       31   /* any of them */
/* NOT a range bit, an ERROR: a wave of a persistent decoder kernel waited for a chunk of the chip-wide tile pool that was never
 * published (the counter block was not zero at launch, or a ring entry was overwritten early) and gave up after ~2^22 polls -- the
 * launch ended instead of hanging, tiles of that call were NOT computed.  The Python binding raises RuntimeError on it. */
#define ADFP_STATUS_POOL_TIMEOUT    32

typedef struct adfp_points {
    int mode;                 /* ADFP_PTS_* */
    long long n_points;       /* P (= n_rays * S in ray mode) */
    const void* pts;          /* [P,3] f64 or f32 (explicit modes) */
    const float* rays_o;      /* [N,3] (ray mode) */
    const float* rays_d;      /* [N,3] */
    const double* z_vals;     /* [N,S] */
    int S;
} adfp_points;

/* ---- info ------------------------------------------------------------------------- */
int adfp_version(void);
/* number of floats of the flat parameter buffer / of the packed image for a decoder kind */
long long adfp_decoder_flat_floats(int kind);
long long adfp_decoder_packed_floats(int kind);
long long adfp_attention_flat_floats(void);
long long adfp_attention_packed_floats(void);
/* bytes of scratch adfp_render_forward / adfp_eval_points need for P points */
size_t adfp_workspace_bytes(long long n_points);

/* ---- layout conversion -------------------------------------------------------------- */
/* [1,32,Z,Y,X] -> [Z,Y,X,32]  (and back, for gradients).  C must be 32. */
int adfp_relayout_grid(const float* src_cm, float* dst_cl, int C, int Z, int Y, int X, void* stream);
int adfp_relayout_grid_back(const float* src_cl, float* dst_cm, int C, int Z, int Y, int X, void* stream);
/* Several grids in ONE launch (a launch costs ~5 us of host time and ~5 us on the device whatever it converts; the Mapper
 * re-materialises three grids per iteration, src/Mapper.py:382-388).  back = 0: [32][V] -> [V][32]; back = 1: the way back.
 * At most ADFP_RELAYOUT_MAX_JOBS jobs; adfp_render_args.relayout_jobs hands the forward conversions to the render call itself. */
typedef struct adfp_relayout_job { const float* src; float* dst; long long voxels; } adfp_relayout_job;
#define ADFP_RELAYOUT_MAX_JOBS 4
int adfp_relayout_grids(int n_jobs, const adfp_relayout_job* jobs /*host*/, int back, void* stream);
/* TSDF volume (any strides; the reference's permuted view of get_tsdf.py:95-97 as it stands) -> its corner-block copy
 * dst[X][Y][Z][8] (8 X Y Z floats, z fastest): block (x, y, z) = the eight values a trilinear lookup with lower corner (x, y, z)
 * blends, v(min(x+dx, X-1), min(y+dy, Y-1), min(z+dz, Z-1)) at index dx + 2 dy + 4 dz.  Built once per volume (the TSDF is
 * static for a run); adfp_tsdf.corner_blocks hands it to the render path.  tsdf->corner_blocks is ignored here. */
int adfp_relayout_tsdf(const adfp_tsdf* tsdf /*host*/, float* dst, void* stream);
/* flat state_dict-order parameters -> packed MFMA image (MLP: decoder.py:91-203) */
int adfp_pack_decoder(int kind, const float* flat, float* packed, void* stream);
/* same parameters -> "H" image (f16 hi/lo halves of every weight), adfp_decoder_packed_h_words(kind) 32-bit words.
 * status (may be NULL): as adfp_scene.status -- ADFP_STATUS_F16_RANGE is raised for a weight with |w| >= 65504. */
long long adfp_decoder_packed_h_words(int kind);
int adfp_pack_decoder_h(int kind, const float* flat, void* packed, int* status, void* stream);
/* -> "T" image for the f16-split backward, adfp_decoder_packed_ht_words(kind) 32-bit words */
long long adfp_decoder_packed_ht_words(int kind);
int adfp_pack_decoder_ht(int kind, const float* flat, void* packed, int* status, void* stream);
long long adfp_attention_packed_h_words(void);
long long adfp_attention_packed_ht_words(void);
int adfp_pack_attention_ht(const float* flat, void* packed, int* status, void* stream);
int adfp_pack_attention_h(const float* flat, void* packed, int* status, void* stream);
/* A split image holds TWO operand layouts back to back: H (v_mfma_f32_32x32x16_f16 order: the training forward, the
 * single-network entries) and G (v_mfma_f32_16x16x32_f16 order: the inference kernels).  adfp_pack_decoder_h / adfp_pack_attention_h
 * write both; this entry writes the chosen part(s) of the same buffer, so that a training iteration re-packs only the H part of
 * a trained network and an inference frame only the G part.  net: ADFP_DEC_LOW / _HIGH / _COLOR or ADFP_NET_ATT; which: bit 0 = H,
 * bit 1 = G.  A consumer must only be handed an image whose part IT reads is current:
 *   H: the high decoder / attention MLP of a training call (adfp_*_train with a state) whose state carries masks_<net>; the low or
 *      the colour decoder whenever it does NOT run inside the fused low + colour launch (stages low / high; the other one on its
 *      exact image; a training state with masks for one of the two only); adfp_decode_single / adfp_attention_rows; (the f16-split
 *      backward reads the separate ht images)
 *   G: the fused low + colour launch = stage colour with both networks on split images, inference or a training state that
 *      carries masks_low AND masks_color; the high decoder / attention MLP of an inference call or of a training call without
 *      masks_<net> (forward = the inference kernel, backward recomputes) */
#define ADFP_NET_ATT 3
#define ADFP_IMAGE_H 1
#define ADFP_IMAGE_G 2
int adfp_pack_split_image(int net, int which, const float* flat, void* packed, int* status, void* stream);
/* Several packed images in ONE launch (a Mapper iteration re-packs four per step: the trained networks' forward parts and their
 * transposed images): job = one image of one network.  format is exactly one of ADFP_IMAGE_H / ADFP_IMAGE_G (that part of the
 * split image buffer `packed`, as adfp_pack_split_image writes it) or ADFP_IMAGE_HT (`packed` = the transposed image of
 * adfp_pack_decoder_ht / adfp_pack_attention_ht).  Same kernels' arithmetic, same status reporting; at most ADFP_PACK_MAX_JOBS jobs. */
#define ADFP_IMAGE_HT 4
#define ADFP_PACK_MAX_JOBS 8
typedef struct adfp_pack_job { int net; int format; const float* flat; void* packed; } adfp_pack_job;
int adfp_pack_images(int n_jobs, const adfp_pack_job* jobs /*host*/, int* status, void* stream);
/* mlp_tsdf parameters (decoder.py:206-258) */
int adfp_pack_attention(const float* flat, float* packed, void* stream);

/* ---- a1: get_rays (src/common.py:254-272) ------------------------------------------- */
/* c2w: [4,4] row-major fp32 on device (only the top 3 rows are read). */
int adfp_get_rays(int H, int W, float fx, float fy, float cx, float cy, const float* c2w,
                  float* rays_o /*[H*W,3]*/, float* rays_d /*[H*W,3]*/, void* stream);

/* ---- a2: get_rays_from_uv (src/common.py:76-91) ----------------------------------------- */
/* Rays through n pixels (pix_i = column, pix_j = row coordinates as float, as get_sample_uv hands them on); c2w [4,4]
 * row-major fp32 on the device.  The backward gives d/d c2w [4,4] (bottom row zero) from the cotangents of the rays:
 * the camera pose of the Tracker and of the Mapper's bundle adjustment reaches the renderer only through this function
 * (src/Tracker.py:97, src/Mapper.py:425).  Either cotangent may be NULL. */
int adfp_rays_from_uv(const float* pix_i, const float* pix_j, int n, float fx, float fy, float cx, float cy, const float* c2w,
                      float* rays_o /*[n,3]*/, float* rays_d /*[n,3]*/, void* stream);
int adfp_rays_from_uv_backward(const float* pix_i, const float* pix_j, int n, float fx, float fy, float cx, float cy,
                               const float* g_rays_o, const float* g_rays_d, float* g_c2w /*[4,4]*/, void* stream);

/* ---- a3: the Mapper's bounding-box pre-filter (src/Mapper.py:438-449) ------------------ */
/* Keeps ray i iff min over axes of max over the two bound planes of (bound - o) / d >= gt_depth
 * (f64 arithmetic on f32 rays, as the reference's f64 `self.bound` promotes it; NaN compares false).
 * bound_dev: [3][2] float64 ON THE DEVICE.  out_index [n_rays] int32 receives the kept ray ids in
 * ascending order (= boolean-mask indexing), *out_count (device int) their number. */
int adfp_prefilter_rays(const float* rays_o, const float* rays_d, const float* gt_depth, int n_rays,
                        const double* bound_dev, int* out_index, int* out_count, void* stream);

/* ---- a4: sampler (src/utils/Renderer.py:134-221) ------------------------------------ */
/* gt_depth may be NULL (then n_surface is ignored, near = 0.01).  t_rand [N,n_samples] is
 * the caller's torch.rand draw when perturb > 0 (Renderer.py:216), else NULL.
 * depth_max: optional device float holding max(gt_depth) over the FULL batch (used by the
 * multi-GPU path so that shards reproduce the single-GPU far clamp, Renderer.py:159/:195);
 * NULL = reduce it here.  scratch: >= 16 bytes of device memory. */
int adfp_sample_rays(const float* rays_o, const float* rays_d, const float* gt_depth, int n_rays,
                     const double bound[3][2], int n_samples, int n_surface, int lindisp,
                     float perturb, const float* t_rand, const float* depth_max,
                     double* z_vals /*[N, S]*/, void* scratch, void* stream);

/* ---- a5..a12: Renderer.eval_points + DF.forward (Renderer.py:27-71, decoder.py:307-353) */
/* raw [P,4] fp32 (rgb, occ), w [P] fp32 (attention weight).
 * flags & ADFP_EVAL_APPLY_BOUND: occ forced to 100 outside scene->bound (Renderer.py:51-64);
 * without it the call is DF.forward alone (what src/utils/Mesher.py:315 calls before applying
 * its own mask). */
#define ADFP_EVAL_APPLY_BOUND 1
int adfp_eval_points(const adfp_scene* scene /*host*/, const adfp_points* pts /*host*/, int stage, int flags,
                     float* raw, float* w, void* workspace, size_t workspace_bytes, void* stream);

/* The same with the state the point backward needs (autograd through Renderer.eval_points / DF.forward: the reference's
 * eval_points is autograd-transparent, src/utils/Renderer.py:27-71).  state = NULL is adfp_eval_points. */
struct adfp_train_state;
int adfp_eval_points_train(const adfp_scene* scene, const adfp_points* pts, int stage, int flags, float* raw, float* w,
                           void* workspace, size_t workspace_bytes, const struct adfp_train_state* state, void* stream);

/* a10 alone: Renderer.sample_grid_tsdf / eval_points_tsdf (Renderer.py:73-107) */
int adfp_sample_tsdf(const adfp_tsdf* tsdf /*host*/, const double tsdf_bnds[3][2],
                     const adfp_points* pts /*host*/, float* out /*[P]*/, void* stream);

/* ---- a13: raw2outputs_nerf_color, occupancy branch (src/common.py:206-251) ------------ */
/* weights may be NULL.  depth/uncertainty are float64 like the reference's. */
int adfp_composite(const float* raw /*[N,S,4]*/, const double* z_vals /*[N,S]*/, int n_rays, int S,
                   double* depth, double* uncertainty, float* color /*[N,3]*/,
                   float* weights /*[N,S] or NULL*/, void* stream);

/* Buffers the backward needs from the forward (training only).  Caller-owned, sized for P points:
 * flags P bytes, list P ints, counter >= 64 bytes (int[0] = length of the in-band list, int[8] = the forward call's f16-range
 * flag: non-zero = the forward was repaired by the f32 fallback, the ReLU masks / layer inputs it left are not valid, and the
 * backward entries then return ZERO gradients for that call -- see adfp_scene.flat_*), att_occ / att_u P floats.
 * Optional, for the f16-split backward (scene->h_* and ->ht_* set; any of them may be NULL = exact backward for that
 * decoder): masks_* = ADFP_TRAIN_MASK_WORDS 32-bit words per point, the ReLU masks of the decoder's five layers;
 * act_* = adfp_train_act_floats(kind) floats per point, the inputs of every layer (position, Fourier features, grid
 * features, h_0..h_4) -- only needed when that decoder's parameter gradient (g_flat_*) will be requested. */
#define ADFP_TRAIN_MASK_WORDS 6
typedef struct adfp_train_state {
    unsigned char* flags;
    int* list;
    int* counter;
    float* att_occ;
    float* att_u;
    unsigned* masks_low;
    unsigned* masks_high;
    unsigned* masks_color;
    float* act_low;
    float* act_high;
    float* act_color;
    unsigned* masks_att;      /* ADFP_TRAIN_ATT_MASK_WORDS words per point (in-band list entry): masks + softmax weights */
    float* act_att;           /* ADFP_TRAIN_ATT_ACT_FLOATS floats per point, only when g_flat_att will be requested */
    /* Debug export, optional (NULL = none), written by the BACKWARD entries: the ReLU decisions the EXACT backward kernels
     * recomputed for a network (they differentiate their own f32 forward, which decides a unit within rounding of zero on its
     * own), in the layout of masks_* / masks_att.  A network that took the f16-split backward used the forward's masks_* and
     * leaves its dbg buffer untouched.  With these a test differentiates the oracle along the SAME piecewise-linear function
     * (oracle/adfp_oracle.py: relu_masks) and holds every gradient element to the forward tolerance. */
    unsigned* dbg_masks_low;
    unsigned* dbg_masks_high;
    unsigned* dbg_masks_color;
    unsigned* dbg_masks_att;
} adfp_train_state;
#define ADFP_TRAIN_ATT_MASK_WORDS 14
#define ADFP_TRAIN_ATT_ACT_FLOATS 416
long long adfp_train_act_floats(int kind);

/* A call whose rays ARE consecutive pixels of one camera frame (Renderer.render_img, src/utils/Renderer.py:278-327, and a rank's
 * contiguous share of it): the call's FIRST launch -- the one that zeroes its device words and packs the weight images it was
 * handed -- also writes the rays of pixels [first, first + n_rays) (src/common.py:254-272, adfp_get_rays' arithmetic) and reduces
 * max(gt_depth) of every ray_batch_size segment of the WHOLE frame (src/utils/Renderer.py:294-313 with :159, :195).  A ray shard
 * then costs no launch for "the rays of the frame" and none for "the maxima of rays it does not hold".
 * `first` = adfp_render_args.depth_max_first_ray; the segment length = depth_max_segment (> 0 required; depth_max must be NULL). */
typedef struct adfp_frame_job {
    const float* c2w;           /* device float[16] (row-major 4x4; rows 0-2 are read): camera-to-world */
    int H, W;
    float fx, fy, cx, cy;
    const float* depth;         /* device float[H*W]: the whole frame's sensor depth; the call's gt_depth = depth + first */
    float* rays_o;              /* [n_rays,3] OUT (caller-owned; the kernels after the first launch read them) */
    float* rays_d;              /* [n_rays,3] OUT */
} adfp_frame_job;

/* ---- a4..a13 in one call: Renderer.render_batch_ray (Renderer.py:110-255) ------------- */
typedef struct adfp_render_args {
    int stage;
    int n_rays;
    int n_samples, n_surface;   /* cfg['rendering'] (configs/df_prior.yaml:93-98) */
    int lindisp;
    float perturb;
    const float* rays_o;        /* [N,3] */
    const float* rays_d;        /* [N,3] */
    const float* gt_depth;      /* [N] or NULL */
    const float* t_rand;        /* [N,n_samples] or NULL */
    const float* depth_max;     /* device float or NULL */
    double* depth;              /* [N]   out */
    double* uncertainty;        /* [N]   out */
    float* color;               /* [N,3] out */
    float* weight;              /* [N,S] out: attention weight (decoder.py:333) */
    double* z_vals;             /* [N,S] out, optional (NULL = keep in workspace) */
    float* raw;                 /* [N,S,4] out, optional */
    void* workspace;
    size_t workspace_bytes;
    const adfp_train_state* state;  /* NULL for inference; else the forward leaves its state here */
    /* > 0: the call's rays are consecutive SEGMENTS of this many rays, each with its own max(gt_depth) for the far clamp and the
     * zero-depth surface range -- what Renderer.render_img's ray batches have (src/utils/Renderer.py:294-313: every 100 000-ray
     * batch clamps `far` with its own maximum) -- so that a whole frame is ONE call with the batched loop's results bit for bit.
     * depth_max, when given, then holds one float per segment; otherwise the maxima are reduced here (at most 48 segments).
     * 0: one maximum for the whole call (render_batch_ray). */
    int depth_max_segment;
    /* With depth_max_segment > 0 and depth_max given: the index, in the segmented batch, of this call's FIRST ray -- the call is a
     * ray shard [first, first + n_rays) of a frame (one GPU's share, attentive_dfprior_amd.dist.render_img_sharded), ray i belongs
     * to segment (first + i) / depth_max_segment, and depth_max holds the maxima of the WHOLE frame's segments.  0 otherwise
     * (non-zero without depth_max or `frame` is ADFP_E_ARG: the call cannot know the maxima of rays it does not hold). */
    int depth_max_first_ray;
    /* Optional: weight images this call's kernels read and that are not packed yet (adfp_pack_images' jobs, host array, at most
     * ADFP_PACK_MAX_JOBS) -- packed by the call's FIRST launch, beside the zero fill of its device words (two launches of ~5 us
     * inside a graph replay otherwise; a Mapper iteration re-packs the trained networks' images every step).  A weight outside
     * the f16 range is reported to scene->status as by adfp_pack_images.  NULL / 0: none. */
    const adfp_pack_job* pack_jobs;
    int n_pack_jobs;
    /* Optional (host struct, NULL = none): the call renders pixels [depth_max_first_ray, + n_rays) of this frame; rays_o / rays_d /
     * gt_depth above are then ignored (the rays are written to frame->rays_o / rays_d, gt_depth = frame->depth + first). */
    const adfp_frame_job* frame;
    /* Optional: feature grids this call's kernels read in channels-last form and that are not converted yet (host array of
     * [32][V] -> [V][32] jobs, at most ADFP_RELAYOUT_MAX_JOBS): converted by extra workgroups of the call's SECOND launch (the
     * sampler: latency-bound, most of the chip idle) -- the decoders, the first readers, are two launches later.  NULL / 0: none. */
    const adfp_relayout_job* relayout_jobs;
    int n_relayout_jobs;
    /* Optional: the Mapper's bounding-box pre-filter (src/Mapper.py:438-449: keep a ray iff min_axis max_side((bound - o) / d) >=
     * gt_depth, in float64) as a job of the call's first launch instead of a launch of its own (adfp_prefilter_mask): that launch
     * writes prefilter_keep[ray] (1 / 0) for every ray of the call, and the call's far clamp is max(gt_depth) over the KEPT rays
     * (adfp_prefilter_mask's depth_max).  All rays are rendered; the caller masks the dropped ones out of loss and gradients
     * (adfp_loss_args.keep, adfp_backward_args.ray_keep).  prefilter_bound: device double[6] = (lo, hi) per axis.  Needs gt_depth,
     * no depth_max, depth_max_segment = 0, no frame.  NULL: none. */
    const double* prefilter_bound;
    unsigned char* prefilter_keep;
} adfp_render_args;

int adfp_render_forward(const adfp_scene* scene /*host*/, const adfp_render_args* args /*host*/, void* stream);

/* ---- a15: backward of render_batch_ray (autograd of src/Mapper.py:457-473) -------------- */
/* Cotangents of (depth, uncertainty, color, weight) -> gradients of the three feature grids
 * (channels-last, converted back by adfp_relayout_grid_back) and of the decoder parameters (flat
 * state_dict order).  Any output pointer may be NULL (= not needed: frozen decoder, lr 0 grid).
 * z_vals, raw and `state` are the ones the forward call wrote.  Every non-NULL output is zeroed and then accumulated:
 * the grid gradients with float atomics (not bitwise reproducible run to run), the parameter gradients atomic-free through
 * per-workgroup partial sums (reproducible).
 * Two implementations per decoder, chosen by what the caller provides: with scene->ht_* and the forward's state->masks_* (and
 * state->act_* when g_flat_* is wanted) the f16-split backward (f16 MFMA, nothing recomputed, grid gradients scattered in
 * sorted order); otherwise -- and always when g_rays_* / g_pts is requested -- the exact f32 backward from scene->w_*.  The
 * attention network likewise (ht_att + state->masks_att [+ act_att], else w_att).
 * g_rays_o / g_rays_d: gradients w.r.t. the rays (p = o + d z; through the trilinear coordinates of the
 * feature grids and the TSDF and through sin(p @ B)) for the Tracker, src/Tracker.py:112-133. */
typedef struct adfp_backward_args {
    int stage;
    int n_rays;
    int S;                       /* samples per ray of the forward call */
    const float* rays_o;
    const float* rays_d;
    const double* z_vals;        /* [N,S] from the forward */
    const float* raw;            /* [N,S,4] from the forward */
    adfp_train_state state;
    const double* g_depth;       /* [N] or NULL */
    const double* g_uncertainty; /* [N] or NULL */
    const float* g_color;        /* [N,3] or NULL */
    const float* g_weight;       /* [N,S] or NULL */
    float* g_grid_low;           /* [Z,Y,X,32] channels-last, or NULL */
    float* g_grid_high;
    float* g_grid_color;
    float* g_flat_low;           /* adfp_decoder_flat_floats(kind) floats, or NULL */
    float* g_flat_high;
    float* g_flat_color;
    float* g_flat_att;           /* adfp_attention_flat_floats() floats, or NULL */
    float* g_rays_o;             /* [N,3] or NULL: camera tracking (src/Tracker.py:112-133) */
    float* g_rays_d;             /* [N,3] or NULL */
    void* workspace;
    size_t workspace_bytes;
    const unsigned char* ray_keep;  /* [N] or NULL: rays flagged 0 (adfp_prefilter_mask) receive no gradient at all */
    int options;                 /* ADFP_BWD_* bits, 0 = defaults */
    /* Optional second lane (NULL = none: everything runs on `stream`, in order).  The spatial sort of the sample points -- nine
     * short, latency-bound launches whose result only k_scatter_sorted, the call's LAST launch, reads -- then runs on
     * `side_stream` BESIDE the backward kernels instead of in front of them: the call records side_events[0] on `stream` when the
     * sort keys exist, makes `side_stream` wait for it and sorts there; before the scatter it records side_events[1] on the lane
     * and makes `stream` wait for it.  While the lane is busy the call's persistent kernels (a whole CU per workgroup) leave
     * ADFP_SIDE_CU_RESERVE compute units free, or the lane's launches would each wait for a whole kernel to retire.  On return
     * `stream` has joined the lane: work queued on `stream` afterwards is ordered after everything, also inside a stream capture
     * (fork and join are captured as graph dependencies).  Caller-owned: a hipStream_t of the same device and two hipEvent_t
     * (timing disabled is fine) that no other call in flight uses. */
    void* side_stream;
    void* side_events[2];
} adfp_backward_args;
/* grid gradients of the f16-split backward through the in-kernel write-combining scatter instead of the sorted scatter
 * (k_bin_keys + radix sort + k_scatter_sorted); same values up to the order of the float atomics */
#define ADFP_BWD_SCATTER_IN_KERNEL 1
/* the caller guarantees that the g_grid_* buffers are all zero on entry (adfp_adam_grids_cl leaves them so): the call does not
 * zero them again */
#define ADFP_BWD_GRIDS_PREZEROED 2
/* weight gradients of the 32-channel decoders through the staged two-kernel path (cotangent blocks written per point, k_outer_h)
 * instead of inside the chain kernel (k_decode_bwd_fused); same values up to the summation order.  What the tests compare the
 * fused kernel against. */
#define ADFP_BWD_STAGED_WGRAD 4
/* weight gradients inside the chain kernel, but with round 3-4's one-wave-per-SIMD kernel (k_decode_bwd_fused: every wave keeps
 * all sixteen gradient blocks) instead of the role-split kernel at two waves per SIMD (k_decode_bwd_roles).  A/B runs. */
#define ADFP_BWD_FUSED_ONE_WAVE 8
size_t adfp_backward_workspace_bytes(long long n_points);
int adfp_render_backward(const adfp_scene* scene /*host*/, const adfp_backward_args* args /*host*/, void* stream);

/* Backward of adfp_eval_points_train: cotangents of raw [P,4] and of the attention weight [P] -> gradients of the grids and
 * decoder parameters (as adfp_render_backward) and of the query points themselves (g_pts [P,3] fp32; through the trilinear
 * coordinates of the feature grids and of the TSDF and through sin(p @ B)).  With ADFP_EVAL_APPLY_BOUND in `flags`, points outside
 * scene->bound pass no occupancy gradient (the forward overwrote their occupancy with 100).  Any output may be NULL. */
typedef struct adfp_points_backward_args {
    int stage;
    int flags;                   /* ADFP_EVAL_* of the forward call */
    adfp_train_state state;      /* the one the forward call filled */
    const float* g_raw;          /* [P,4] or NULL */
    const float* g_w;            /* [P] or NULL */
    float* g_grid_low;           /* [Z,Y,X,32] channels-last, or NULL */
    float* g_grid_high;
    float* g_grid_color;
    float* g_flat_low;           /* adfp_decoder_flat_floats(kind) floats, or NULL */
    float* g_flat_high;
    float* g_flat_color;
    float* g_flat_att;
    float* g_pts;                /* [P,3] or NULL */
    void* workspace;             /* adfp_backward_workspace_bytes(P) */
    size_t workspace_bytes;
    int options;                 /* ADFP_BWD_* bits, 0 = defaults */
} adfp_points_backward_args;
int adfp_eval_points_backward(const adfp_scene* scene /*host*/, const adfp_points* pts /*host*/, const adfp_points_backward_args* args /*host*/,
                              void* stream);

/* ---- TSDF fusion of one RGB-D frame (src/fusion.py:69-142 CUDA kernel, launch :226-251) ---- */
/* tsdf / weight / color: device volumes in the reference's physical order [X][Y][Z] (Z fastest), updated in
 * place; color may be NULL.  origin, cam_intr (3x3) and cam_pose (4x4 camera-to-world, already in the
 * OpenCV convention of get_tsdf.py:79-80) are HOST arrays; color_im is the packed b*65536+g*256+r image
 * (src/fusion.py:223), depth_im the depth image, both [H,W] fp32 on the device. */
int adfp_tsdf_integrate(float* tsdf, float* weight, float* color, int dim_x, int dim_y, int dim_z, const float origin[3],
                        float voxel_size, const float cam_intr[9], const float cam_pose[16], const float* color_im,
                        const float* depth_im, int im_h, int im_w, float trunc_margin, float obs_weight, void* stream);

/* ---- Mapper bookkeeping on the device ----------------------------------- */
/* Frustum feature selection, src/Mapper.py:90-158: which points of an X x Y x Z feature grid (point
 * coordinates = linspace over `bound` per axis) project into the current depth image in front of the sensed
 * surface (+0.5 m), plus the ball of radius 0.5 m around the camera centre.  The reference does this on the
 * host with numpy and cv2.remap (bilinear, 1/32-pixel map rounding, constant-0 border -- restated here).
 * c2w / w2c (= inverse, computed by the caller) are HOST 4x4 row-major fp32; depth is the [H,W] fp32 image on
 * the device; sampled: X*Y*Z floats of device workspace; scratch: 4 bytes of device workspace.
 * mask: X*Y*Z bytes written in the grid tensor's [Z][Y][X] order (the permute(2,1,0) of src/Mapper.py:345). */
int adfp_frustum_mask(int X, int Y, int Z, const double bound[3][2], const float c2w[16], const float w2c[16],
                      double fx, double fy, double cx, double cy, int H, int W, const float* depth,
                      float* sampled, unsigned* scratch, unsigned char* mask, void* stream);
/* One torch.optim.Adam step (amsgrad off, weight_decay 0; src/Mapper.py:374-378, :473) on the masked
 * voxels of a channel-major grid [channels][nvox], in place -- instead of the reference's compact copy
 * `val[mask]` that is index_put into the grid before and after every iteration (src/Mapper.py:347-361,
 * :382-388, :476-482).  exp_avg / exp_avg_sq have the grid's shape and must be zero before step 1; elements
 * outside the mask (mask == NULL: none) are not touched.  step counts from 1. */
int adfp_masked_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const unsigned char* mask,
                     long long nvox, int channels, float lr, float beta1, float beta2, float eps, int step, void* stream);

/* ---- One Mapper iteration as a fixed, sync-free kernel sequence (src/Mapper.py:438-473) ------------------------------- */
/* The bounding-box pre-filter of adfp_prefilter_rays as a per-ray keep flag (1 = kept) plus *depth_max = the max sensor
 * depth of the KEPT rays (device float, feeds adfp_render_args.depth_max).  The reference compacts the batch with boolean
 * indexing (a device sync, a data-dependent batch size); rendering the dropped rays too and masking them out of the loss
 * (adfp_loss_args.keep, adfp_backward_args.ray_keep) gives the kept rays identical outputs and gradients, because a ray
 * sees the rest of its batch only through max(gt_depth) (src/utils/Renderer.py:159, :195). */
int adfp_prefilter_mask(const float* rays_o, const float* rays_d, const float* gt_depth, int n_rays, const double* bound_dev,
                        unsigned char* keep /*[N]*/, float* depth_max /*device float*/, void* stream);
/* The Mapper's loss (src/Mapper.py:457-469) and its cotangents w.r.t. the renderer's outputs:
 *   sum_{gt_depth > 0} |gt_depth - depth|  [+ sum |weight - 1| when `warmup`]  [+ w_color_loss * sum |gt_color - color| in stage color]
 * g_* = d loss / d output (sign functions, 0 at 0 like torch.abs's backward), zero for rays with keep == 0.
 * loss (device double, may be NULL) is ACCUMULATED: zero it first. */
typedef struct adfp_loss_args {
    int n_rays, S;
    int stage;                   /* ADFP_STAGE_*; the colour term exists in stage color only */
    int warmup;                  /* src/Mapper.py:459: idx <= 1 and the 5 iterations after the low stage */
    float w_color_loss;          /* configs/df_prior.yaml:56 */
    const double* depth;         /* [N]   */
    const float* color;          /* [N,3] (stage color) */
    const float* weight;         /* [N,S] attention weight (warm-up) */
    const float* gt_depth;       /* [N]   */
    const float* gt_color;       /* [N,3] (stage color) */
    const unsigned char* keep;   /* [N] or NULL */
    double* loss;                /* device double or NULL */
    double* g_depth;             /* [N]   out */
    float* g_color;              /* [N,3] out (stage color; may be NULL otherwise) */
    float* g_weight;             /* [N,S] out (warm-up; may be NULL otherwise) */
} adfp_loss_args;
int adfp_mapper_loss(const adfp_loss_args* args /*host*/, void* stream);
/* The same kernel doing three launches' work (inside a graph replay a launch costs ~5 us whatever it does):
 *  - the loss is WRITTEN, not accumulated (no zero fill first): per-workgroup partial sums go to scratch + 8 and the workgroup
 *    that draws the last ticket adds them up in order (reproducible, unlike the atomics of adfp_mapper_loss).  scratch: device
 *    memory, 8-byte aligned, adfp_mapper_loss_scratch_bytes(n_rays) bytes, whose first int is ZERO before the first call (the
 *    kernel leaves it zero);
 *  - adfp_adam_prep's work (below: same arguments, n_groups may be 0) is done by the first workgroup on the side.
 * With n_rays == 0 nothing is launched: zero the loss and call adfp_adam_prep yourself. */
size_t adfp_mapper_loss_scratch_bytes(int n_rays);
int adfp_mapper_loss_step(const adfp_loss_args* args /*host*/, void* scratch, size_t scratch_bytes, int* steps, float* derived, int n_groups,
                          const float* lr /*host*/, float beta1, float beta2, const int* skip_flag, void* stream);
/* torch.optim.Adam's step counters and bias corrections on the device, for n_groups <= 8 parameter groups in ONE launch:
 * for every group g with lr[g] >= 0:  steps[g] += 1,  derived[2g] = lr[g] / (1 - beta1^steps[g]),  derived[2g+1] =
 * sqrt(1 - beta2^steps[g])  (double arithmetic, one rounding, like the python floats of torch.optim); a negative lr[g]
 * = the group has no gradient in this iteration and does not step.  lr is a HOST array. */
/* skip_flag (device int, may be NULL): when *skip_flag != 0 at run time no group steps and `derived` is zeroed, which makes
 * adfp_masked_adam_dev / _multi leave parameters and moments untouched -- the Mapper iteration hands over the forward call's
 * f16-range flag (adfp_train_state.counter + 8) so that an iteration whose gradients are not valid changes nothing. */
int adfp_adam_prep(int* steps /*device int[n]*/, float* derived /*device float[n][2]*/, int n_groups, const float* lr /*host*/,
                   float beta1, float beta2, const int* skip_flag, void* stream);
/* adfp_masked_adam with the step-dependent scalars read from `derived`: no host value changes from step to step, so
 * the call can be replayed from a HIP graph. */
int adfp_masked_adam_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const unsigned char* mask,
                         long long nvox, int channels, float beta1, float beta2, float eps, const float* derived, void* stream);
/* The same for up to 8 parameter groups in ONE launch (a Mapper iteration steps three grids and two networks). */
typedef struct adfp_adam_group {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    const unsigned char* mask;   /* [nvox] or NULL */
    long long nvox; int channels;
    const float* derived;        /* this group's {step size, sqrt(bias correction 2)} from adfp_adam_prep */
} adfp_adam_group;
int adfp_masked_adam_multi(int n_groups, const adfp_adam_group* groups /*host*/, float beta1, float beta2, float eps, void* stream);
/* The same step on feature grids whose optimiser state lives in the KERNELS' layout: gradient (as adfp_render_backward writes it),
 * both moments and a shadow copy of the parameters are channels-last [nvox][32]; the reference-layout grid [32][nvox] that the
 * rest of the system sees (the Tracker reads it from another process, src/Tracker.py:144-147) is written through.  One launch
 * replaces, per grid and iteration, the forward's re-layout of the updated grid, the backward's re-layout of the gradient and
 * the zeroing of the gradient buffer: every gradient element is set to zero as it is consumed (masked or not), so the buffer
 * can go straight into the next adfp_render_backward with ADFP_BWD_GRIDS_PREZEROED.  A skipped step (derived == {0, 0},
 * adfp_adam_prep's skip_flag) still zeroes the gradient.  Up to 8 grids per launch. */
typedef struct adfp_adam_cl_group {
    float* param_cl;             /* [nvox][32] shadow of the grid the render kernels read */
    float* param_cm;             /* [32][nvox] the reference-layout grid, written through on the masked voxels */
    float* grad_cl;              /* [nvox][32], consumed and zeroed */
    float* exp_avg_cl; float* exp_avg_sq_cl;
    const unsigned char* mask;   /* [nvox] or NULL */
    long long nvox;
    const float* derived;        /* {step size, sqrt(bias correction 2)} from adfp_adam_prep */
} adfp_adam_cl_group;
int adfp_adam_grids_cl(int n_groups, const adfp_adam_cl_group* groups /*host*/, float beta1, float beta2, float eps, void* stream);
/* adfp_adam_grids_cl and adfp_masked_adam_multi in ONE launch: every parameter group a Mapper iteration steps (either list may be empty) */
int adfp_adam_step(int n_cl_groups, const adfp_adam_cl_group* cl_groups /*host*/, int n_groups, const adfp_adam_group* groups /*host*/,
                   float beta1, float beta2, float eps, void* stream);

/* ---- One Tracker iteration as a fixed, sync-free kernel sequence (src/Tracker.py:75-134) ---------------------------------- */
/* Camera tensor (quaternion r, i, j, k + translation, 7 floats ON THE DEVICE) -> camera-to-world [4,4] row-major, and the
 * cotangent of c2w (rows 0-2) -> the cotangent of the 7 parameters: get_camera_from_tensor / quad2rotation of
 * src/common.py:139-178 and their autograd.  The pose the Tracker optimises reaches the renderer only through these. */
int adfp_camera_from_tensor(const float* cam /*[7] device*/, float* c2w /*[16] device*/, void* stream);
int adfp_camera_from_tensor_backward(const float* cam, const float* g_c2w /*[16]*/, float* g_cam /*[7]*/, void* stream);
/* The n sampled pixels of get_sample_uv (src/common.py:94-124): idx[t] indexes the window [H0,H1) x [W0,W1) in row-major
 * order (the caller's ONE torch.randint draw); out: pixel coordinates as floats (pix_i = column, pix_j = row, what
 * adfp_rays_from_uv takes), sensor depth [n] and colour [n,3] gathered from the [H,W] / [H,W,3] fp32 images. */
int adfp_select_pixels(const long long* idx /*[n] int64 device*/, int n, int H0, int H1, int W0, int W1, int H, int W,
                       const float* depth_img, const float* color_img, float* pix_i, float* pix_j, float* gt_depth, float* gt_color,
                       void* stream);
/* The Mapper's ray batch of one iteration (src/Mapper.py:421-436: get_samples per keyframe of the optimisation window, then four
 * torch.cat) in ONE launch: frame f contributes n rays -- pixel idx[t] of the window like adfp_select_pixels, its ray like
 * adfp_rays_from_uv (bit for bit), sensor depth and colour -- at rows [f n, f n + n) of the outputs.  The draws stay the caller's (one
 * torch.randint per frame: the reference's index stream).  c2w: device [4,4] row-major fp32, or NULL and the pose in c2w_host (rows
 * 0-2 of the matrix, row-major).  No gradient towards the poses (bundle adjustment takes the per-frame entries). */
#define ADFP_KEYFRAMES_MAX 16
typedef struct adfp_keyframe {
    const long long* idx;      /* [n] int64, device */
    const float* c2w;          /* device, or NULL */
    float c2w_host[12];
    const float* depth_img;    /* [H,W] fp32 */
    const float* color_img;    /* [H,W,3] fp32 */
} adfp_keyframe;
int adfp_sample_keyframes(int n_frames, const adfp_keyframe* frames /*host*/, int n, int H0, int H1, int W0, int W1, int H, int W,
                          float fx, float fy, float cx, float cy, float* rays_o /*[n_frames n,3]*/, float* rays_d, float* gt_depth /*[n_frames n]*/,
                          float* gt_color /*[n_frames n,3]*/, void* stream);
/* The head and the tail of a Tracker iteration as ONE launch each (a launch costs ~5 us inside a graph replay whatever it does, and
 * the chain below is seven of them).
 * adfp_tracker_head = adfp_camera_from_tensor + adfp_select_pixels + adfp_rays_from_uv + adfp_prefilter_mask, same arithmetic:
 *   cam [7] -> c2w [16]; the n drawn pixels -> pix_i / pix_j / gt_depth / gt_color, their rays, the bounding-box keep flags and
 *   the largest sensor depth of the kept rays.
 * adfp_tracker_tail = adfp_rays_from_uv_backward + adfp_camera_from_tensor_backward [+ adfp_adam_prep + adfp_masked_adam_multi on
 *   the pose's parameter groups + adfp_track_keep_best]: ray cotangents -> g_c2w [16] -> g_cam [7]; with step != 0 the Adam step of
 *   torch.optim.Adam on cam (groups: n_groups = 1 -> the 7 parameters with lr[0]; 2 -> translation cam[4..7) with lr[0], quaternion
 *   cam[0..4) with lr[1], src/Tracker.py:219-229) and the running best pose -- kept BEFORE the step when n_groups = 2, after it
 *   when n_groups = 1 (the reference rebuilds / clones its camera tensor at those points, src/Tracker.py:236-263).  steps [n_groups]
 *   int / derived [n_groups][2] float as adfp_adam_prep; skip_flag as there. */
typedef struct adfp_tracker_head_args {
    const float* cam; float* c2w;
    const long long* idx; int n; int H0, H1, W0, W1, H, W;
    const float* depth_img; const float* color_img;
    float fx, fy, cx, cy;
    const double* bound;                     /* device [6] */
    float* pix_i; float* pix_j; float* gt_depth; float* gt_color; float* rays_o; float* rays_d;
    unsigned char* keep; float* depth_max;
} adfp_tracker_head_args;
int adfp_tracker_head(const adfp_tracker_head_args* args /*host*/, void* stream);
typedef struct adfp_tracker_tail_args {
    const float* pix_i; const float* pix_j; int n; float fx, fy, cx, cy;
    const float* g_rays_o; const float* g_rays_d;
    float* cam; float* g_c2w; float* g_cam;
    int step;                                /* 0: gradients only */
    float* exp_avg; float* exp_avg_sq;       /* [7] each */
    int* steps; float* derived; int n_groups; float lr[2]; float beta1, beta2, eps; const int* skip_flag;
    const double* loss; double* best_loss; float* best_cam;
} adfp_tracker_tail_args;
int adfp_tracker_tail(const adfp_tracker_tail_args* args /*host*/, void* stream);
/* The tracking loss (src/Tracker.py:115-129) and its cotangents:
 *   tmp = |gt_depth - depth| / sqrt(uncertainty + 1e-10)        (float64, uncertainty detached)
 *   mask = keep & (gt_depth > 0) [& tmp < 10 median(tmp over the kept rays) when handle_dynamic]
 *   loss = sum_mask tmp + w_color_loss sum_mask |gt_color - color|
 * torch.median's lower-middle element; at most 8192 rays (one workgroup).  loss (device double) is WRITTEN, not accumulated. */
typedef struct adfp_track_loss_args {
    int n_rays;
    int handle_dynamic;          /* configs/df_prior.yaml:28 */
    float w_color_loss;          /* :31 */
    const double* depth;         /* [N] */
    const double* uncertainty;   /* [N] */
    const float* color;          /* [N,3] */
    const float* gt_depth;       /* [N] */
    const float* gt_color;       /* [N,3] */
    const unsigned char* keep;   /* [N] or NULL: the bounding-box pre-filter of :100-109 as a keep flag (adfp_prefilter_mask) */
    double* loss;                /* device double or NULL */
    double* g_depth;             /* [N] out */
    float* g_color;              /* [N,3] out */
} adfp_track_loss_args;
int adfp_tracker_loss(const adfp_track_loss_args* args /*host*/, void* stream);
/* The iteration loop's running best (src/Tracker.py:261-263): if *loss < *best_loss then *best_loss = *loss and best_cam[0..7) =
 * cam[0..7); all four on the device, NaN never wins.  Start best_loss at +inf (the reference's 1e10). */
int adfp_track_keep_best(const double* loss, const float* cam, double* best_loss, float* best_cam, void* stream);

/* Stable radix sort of n (key, value) int pairs by the low key_bits bits of the (non-negative) keys, in place (key_tmp / val_tmp:
 * n ints each).  What the f16-split backward orders the sample points with (by grid cell, adfp_sort.h); exported for testing. */
size_t adfp_sort_workspace_bytes(long long n);
int adfp_sort_pairs(int* key, int* val, int* key_tmp, int* val_tmp, long long n, int key_bits, void* workspace, size_t workspace_bytes,
                    void* stream);

/* Sort keys that bring a large batch of rays in INCOHERENT order (a random subset of several images' pixels) into a spatially
 * coherent one before it is rendered: key[i] = (Morton code of the ray origin's cell, 2 bits per axis) << 24 | Morton code of the
 * cell of the ray's surface point o + d * gt_depth (gt_depth <= 0 / NULL: depth 1), 8 bits per axis, both inside tsdf_bnds; val[i]
 * = i.  Sorted with adfp_sort_pairs (30 key bits) the rays of one camera that look at one 1/256 cell of the scene become
 * neighbours, and the TSDF stage's wave-wide loads and the decoders' grid gathers find their lines and pages shared again
 * (1024^3 volume, 131 072 random rays x 128 samples: 5.49 -> 4.93 ms per batch; rays are independent units, so the rendered values
 * do not depend on the order).  tsdf_bnds is a HOST array. */
int adfp_ray_sort_keys(const float* rays_o, const float* rays_d, const float* gt_depth /*or NULL*/, int n_rays, const double tsdf_bnds[3][2],
                       int* key, int* val, void* stream);
/* Is a batch in a coherent order?  Looks at up to 2048 evenly spread pairs of CONSECUTIVE rays and writes to verdict[0] how many
 * of them have surface points further apart than `far_distance` (a few TSDF voxels), and to verdict[1] the number of pairs
 * looked at.  verdict: two ints in device-visible memory (pinned host memory: the host reads it later without a sync). */
int adfp_ray_order_probe(const float* rays_o, const float* rays_d, const float* gt_depth /*or NULL*/, int n_rays, float far_distance,
                         int* verdict, void* stream);

/* ---- multi-GPU render (new functionality; the reference has no distributed code): the send / receive side of ONE all-gather ----
 * A sharded render returns several per-ray arrays (depth f64, uncertainty f64, colour 3 x f32 ... = 28 B per ray).  pack writes
 * the rank's `rows` rows of all of them interleaved into dst [rows][sum words] (the send buffer: the rank's slot of the gather
 * buffer, or a buffer padded to the largest shard); unpack reads the gathered [world][pad][sum words] buffer and writes every
 * array contiguous in ray order, rank r contributing rows_per_rank[r] <= pad rows.  Row widths in 4-byte words; at most
 * ADFP_GATHER_MAX arrays and ADFP_GATHER_MAX_RANKS ranks; src / dst / words / rows_per_rank are HOST arrays. */
#define ADFP_GATHER_MAX 8
#define ADFP_GATHER_MAX_RANKS 64
int adfp_gather_pack(int n_arrays, const void* const* src, const int* words, long long rows, void* dst, void* stream);
int adfp_gather_unpack(int n_arrays, void* const* dst, const int* words, int world, long long pad, const long long* rows_per_rank,
                       const void* gathered, void* stream);

/* Per-stage timing hook for bench.py: runs ONLY the TSDF trilerp + band-mask kernel (a10).
 * w may be NULL (that is the render path's launch: there the LOW decoder writes w = 1). */
int adfp_tsdf_stage(const adfp_scene* scene, const adfp_points* pts, unsigned char* flags,
                    int* list, float* att_u, float* w, int* counter, void* stream);

/* Per-stage timing hook for bench.py: runs ONLY one decoder kernel (kind = ADFP_DEC_LOW or ADFP_DEC_COLOR, or
 * ADFP_DEC_LOW_COLOR = the fused low + colour launch that stage color uses with f16-split images) over every point and
 * writes its output channel(s) of raw.  tile_counter: one device int the fused launch may use for its chip-wide tile tail (as it
 * does inside adfp_render_forward, where the word lives in the workspace; zeroed here before the launch), or NULL = fixed split. */
#define ADFP_DEC_LOW_COLOR 3
int adfp_decode_stage(const adfp_scene* scene, const adfp_points* pts, int kind, float* raw, float* w, int* tile_counter, void* stream);

/* ---- one sub-network alone (reference: the public modules `decoders.low_decoder / high_decoder / color_decoder / mlp`) ---- */
/* MLP.forward(p, c_grid) of src/conv_onet/models/decoder.py:177-203 for one decoder kind, no bound rule, no band logic.
 * LOW: out4 is [P,4], the value lands in channel 3 (channels 0-2 untouched); COLOR: out4 is [P,4], all four outputs of the
 * colour head (the renderer discards the fourth, decoder.py:351-352); HIGH (concat_feature: own grid + low grid): out4 is [P]. */
int adfp_decode_single(const adfp_scene* scene /*host*/, const adfp_points* pts /*host*/, int kind, float* out4, void* stream);
/* mlp_tsdf.forward of decoder.py:240-258 on explicit rows: occ [n] and the trilinear TSDF value tsdf_val [n] (adfp_sample_tsdf)
 * -> fused occupancy out4[4 i + 3] and the attention weight w[i].  scratch_u: n floats (inv_tsdf of the rows). */
int adfp_attention_rows(const adfp_scene* scene /*host*/, const float* occ, const float* tsdf_val, long long n, float* out4 /*[n,4]*/,
                        float* w /*[n]*/, float* scratch_u, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADFP_H */
