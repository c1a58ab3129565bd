"""ctypes binding of libadfp.so (include/adfp.h).

The product path has no CPU fallback: if the HIP library is missing, ``lib()`` raises and
every compute entry point of this package fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ADFP_LIB_PATH') or os.path.join(_HERE, 'libadfp.so')   # override: kernel A/B builds

ABI_VERSION = 131                 # ADFP_VERSION of include/adfp.h this binding was written against
STATUS_F16_RANGE = 31              # ADFP_STATUS_F16_RANGE: any of the bits below
STATUS_RANGE_BITS = {'low': 1, 'high': 2, 'color': 4, 'att': 8, 'bwd': 16}      # ADFP_STATUS_F16_RANGE_<net>
BWD_SCATTER_IN_KERNEL = 1        # ADFP_BWD_SCATTER_IN_KERNEL
BWD_GRIDS_PREZEROED = 2          # ADFP_BWD_GRIDS_PREZEROED
BWD_STAGED_WGRAD = 4             # ADFP_BWD_STAGED_WGRAD
BWD_FUSED_ONE_WAVE = 8           # ADFP_BWD_FUSED_ONE_WAVE
STAGE = {'low': 0, 'high': 1, 'color': 2}
DEC_KIND = {'low': 0, 'high': 1, 'color': 2}
NET_ID = {'low': 0, 'high': 1, 'color': 2, 'att': 3}      # adfp_pack_split_image's `net`
IMAGE_H, IMAGE_G, IMAGE_HT = 1, 2, 4        # ADFP_IMAGE_*
PTS_RAYS, PTS_F64, PTS_F32 = 0, 1, 2

ERRORS = {-1: 'ADFP_E_ARG (null pointer / bad size)',
          -2: 'ADFP_E_UNSUPPORTED',
          -3: 'ADFP_E_WORKSPACE (workspace too small)'}


class AdfpGrid(C.Structure):
    _fields_ = [('data', C.c_void_p), ('Z', C.c_int), ('Y', C.c_int), ('X', C.c_int)]


class AdfpTsdf(C.Structure):
    _fields_ = [('data', C.c_void_p), ('Z', C.c_int), ('Y', C.c_int), ('X', C.c_int),
                ('sZ', C.c_longlong), ('sY', C.c_longlong), ('sX', C.c_longlong), ('corner_blocks', C.c_void_p)]


class AdfpPackJob(C.Structure):
    _fields_ = [('net', C.c_int), ('format', C.c_int), ('flat', C.c_void_p), ('packed', C.c_void_p)]


class AdfpRelayoutJob(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('voxels', C.c_longlong)]


RELAYOUT_MAX_JOBS = 4


class AdfpTrackerHeadArgs(C.Structure):
    _fields_ = [('cam', C.c_void_p), ('c2w', C.c_void_p), ('idx', C.c_void_p), ('n', C.c_int),
                ('H0', C.c_int), ('H1', C.c_int), ('W0', C.c_int), ('W1', C.c_int), ('H', C.c_int), ('W', C.c_int),
                ('depth_img', C.c_void_p), ('color_img', C.c_void_p),
                ('fx', C.c_float), ('fy', C.c_float), ('cx', C.c_float), ('cy', C.c_float), ('bound', C.c_void_p),
                ('pix_i', C.c_void_p), ('pix_j', C.c_void_p), ('gt_depth', C.c_void_p), ('gt_color', C.c_void_p),
                ('rays_o', C.c_void_p), ('rays_d', C.c_void_p), ('keep', C.c_void_p), ('depth_max', C.c_void_p)]


class AdfpTrackerTailArgs(C.Structure):
    _fields_ = [('pix_i', C.c_void_p), ('pix_j', C.c_void_p), ('n', C.c_int),
                ('fx', C.c_float), ('fy', C.c_float), ('cx', C.c_float), ('cy', C.c_float),
                ('g_rays_o', C.c_void_p), ('g_rays_d', C.c_void_p), ('cam', C.c_void_p), ('g_c2w', C.c_void_p), ('g_cam', C.c_void_p),
                ('step', C.c_int), ('exp_avg', C.c_void_p), ('exp_avg_sq', C.c_void_p),
                ('steps', C.c_void_p), ('derived', C.c_void_p), ('n_groups', C.c_int), ('lr', C.c_float * 2),
                ('beta1', C.c_float), ('beta2', C.c_float), ('eps', C.c_float), ('skip_flag', C.c_void_p),
                ('loss', C.c_void_p), ('best_loss', C.c_void_p), ('best_cam', C.c_void_p)]


class AdfpKeyframe(C.Structure):
    _fields_ = [('idx', C.c_void_p), ('c2w', C.c_void_p), ('c2w_host', C.c_float * 12), ('depth_img', C.c_void_p), ('color_img', C.c_void_p)]


KEYFRAMES_MAX = 16


class AdfpScene(C.Structure):
    _fields_ = [('bound', (C.c_double * 2) * 3), ('tsdf_bnds', (C.c_double * 2) * 3),
                ('low', AdfpGrid), ('high', AdfpGrid), ('color', AdfpGrid), ('tsdf', AdfpTsdf),
                ('w_low', C.c_void_p), ('w_high', C.c_void_p), ('w_color', C.c_void_p), ('w_att', C.c_void_p),
                ('h_low', C.c_void_p), ('h_high', C.c_void_p), ('h_color', C.c_void_p), ('h_att', C.c_void_p),
                ('ht_low', C.c_void_p), ('ht_high', C.c_void_p), ('ht_color', C.c_void_p), ('ht_att', C.c_void_p),
                ('flat_low', C.c_void_p), ('flat_high', C.c_void_p), ('flat_color', C.c_void_p), ('flat_att', C.c_void_p),
                ('status', C.c_void_p)]


class AdfpPoints(C.Structure):
    _fields_ = [('mode', C.c_int), ('n_points', C.c_longlong), ('pts', C.c_void_p),
                ('rays_o', C.c_void_p), ('rays_d', C.c_void_p), ('z_vals', C.c_void_p), ('S', C.c_int)]


class AdfpTrainState(C.Structure):
    _fields_ = [('flags', C.c_void_p), ('list', C.c_void_p), ('counter', C.c_void_p),
                ('att_occ', C.c_void_p), ('att_u', C.c_void_p),
                ('masks_low', C.c_void_p), ('masks_high', C.c_void_p), ('masks_color', C.c_void_p),
                ('act_low', C.c_void_p), ('act_high', C.c_void_p), ('act_color', C.c_void_p),
                ('masks_att', C.c_void_p), ('act_att', C.c_void_p),
                ('dbg_masks_low', C.c_void_p), ('dbg_masks_high', C.c_void_p), ('dbg_masks_color', C.c_void_p), ('dbg_masks_att', C.c_void_p)]


TRAIN_MASK_WORDS = 6              # ADFP_TRAIN_MASK_WORDS
TRAIN_ATT_MASK_WORDS = 14         # ADFP_TRAIN_ATT_MASK_WORDS
TRAIN_ATT_ACT_FLOATS = 416        # ADFP_TRAIN_ATT_ACT_FLOATS


class AdfpAdamGroup(C.Structure):
    _fields_ = [('param', C.c_void_p), ('grad', C.c_void_p), ('exp_avg', C.c_void_p), ('exp_avg_sq', C.c_void_p),
                ('mask', C.c_void_p), ('nvox', C.c_longlong), ('channels', C.c_int), ('derived', C.c_void_p)]


class AdfpAdamClGroup(C.Structure):
    _fields_ = [('param_cl', C.c_void_p), ('param_cm', C.c_void_p), ('grad_cl', C.c_void_p), ('exp_avg_cl', C.c_void_p),
                ('exp_avg_sq_cl', C.c_void_p), ('mask', C.c_void_p), ('nvox', C.c_longlong), ('derived', C.c_void_p)]


class AdfpFrameJob(C.Structure):
    _fields_ = [('c2w', C.c_void_p), ('H', C.c_int), ('W', C.c_int), ('fx', C.c_float), ('fy', C.c_float), ('cx', C.c_float), ('cy', C.c_float),
                ('depth', C.c_void_p), ('rays_o', C.c_void_p), ('rays_d', C.c_void_p)]


class AdfpRenderArgs(C.Structure):
    _fields_ = [('stage', C.c_int), ('n_rays', C.c_int), ('n_samples', C.c_int), ('n_surface', C.c_int),
                ('lindisp', C.c_int), ('perturb', C.c_float),
                ('rays_o', C.c_void_p), ('rays_d', C.c_void_p), ('gt_depth', C.c_void_p),
                ('t_rand', C.c_void_p), ('depth_max', C.c_void_p),
                ('depth', C.c_void_p), ('uncertainty', C.c_void_p), ('color', C.c_void_p),
                ('weight', C.c_void_p), ('z_vals', C.c_void_p), ('raw', C.c_void_p),
                ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t), ('state', C.POINTER(AdfpTrainState)),
                ('depth_max_segment', C.c_int), ('depth_max_first_ray', C.c_int),
                ('pack_jobs', C.c_void_p), ('n_pack_jobs', C.c_int), ('frame', C.POINTER(AdfpFrameJob)),
                ('relayout_jobs', C.c_void_p), ('n_relayout_jobs', C.c_int), ('prefilter_bound', C.c_void_p), ('prefilter_keep', C.c_void_p)]


class AdfpBackwardArgs(C.Structure):
    _fields_ = [('stage', C.c_int), ('n_rays', C.c_int), ('S', C.c_int),
                ('rays_o', C.c_void_p), ('rays_d', C.c_void_p), ('z_vals', C.c_void_p), ('raw', C.c_void_p),
                ('state', AdfpTrainState),
                ('g_depth', C.c_void_p), ('g_uncertainty', C.c_void_p), ('g_color', C.c_void_p), ('g_weight', C.c_void_p),
                ('g_grid_low', C.c_void_p), ('g_grid_high', C.c_void_p), ('g_grid_color', C.c_void_p),
                ('g_flat_low', C.c_void_p), ('g_flat_high', C.c_void_p), ('g_flat_color', C.c_void_p),
                ('g_flat_att', C.c_void_p), ('g_rays_o', C.c_void_p), ('g_rays_d', C.c_void_p),
                ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t), ('ray_keep', C.c_void_p), ('options', C.c_int),
                ('side_stream', C.c_void_p), ('side_events', C.c_void_p * 2)]


class AdfpPointsBackwardArgs(C.Structure):
    _fields_ = [('stage', C.c_int), ('flags', C.c_int), ('state', AdfpTrainState), ('g_raw', C.c_void_p), ('g_w', C.c_void_p),
                ('g_grid_low', C.c_void_p), ('g_grid_high', C.c_void_p), ('g_grid_color', C.c_void_p),
                ('g_flat_low', C.c_void_p), ('g_flat_high', C.c_void_p), ('g_flat_color', C.c_void_p), ('g_flat_att', C.c_void_p),
                ('g_pts', C.c_void_p), ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t), ('options', C.c_int)]


class AdfpLossArgs(C.Structure):
    _fields_ = [('n_rays', C.c_int), ('S', C.c_int), ('stage', C.c_int), ('warmup', C.c_int), ('w_color_loss', C.c_float),
                ('depth', C.c_void_p), ('color', C.c_void_p), ('weight', C.c_void_p), ('gt_depth', C.c_void_p),
                ('gt_color', C.c_void_p), ('keep', C.c_void_p), ('loss', C.c_void_p), ('g_depth', C.c_void_p),
                ('g_color', C.c_void_p), ('g_weight', C.c_void_p)]


class AdfpTrackLossArgs(C.Structure):
    _fields_ = [('n_rays', C.c_int), ('handle_dynamic', C.c_int), ('w_color_loss', C.c_float),
                ('depth', C.c_void_p), ('uncertainty', C.c_void_p), ('color', C.c_void_p), ('gt_depth', C.c_void_p),
                ('gt_color', C.c_void_p), ('keep', C.c_void_p), ('loss', C.c_void_p), ('g_depth', C.c_void_p), ('g_color', C.c_void_p)]


Bound = (C.c_double * 2) * 3

# every symbol include/adfp.h declares: (name, restype, argtypes)
SYMBOLS = [
    ('adfp_version', C.c_int, []),
    ('adfp_decoder_flat_floats', C.c_longlong, [C.c_int]),
    ('adfp_decoder_packed_floats', C.c_longlong, [C.c_int]),
    ('adfp_attention_flat_floats', C.c_longlong, []),
    ('adfp_attention_packed_floats', C.c_longlong, []),
    ('adfp_workspace_bytes', C.c_size_t, [C.c_longlong]),
    ('adfp_relayout_grid', C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ('adfp_relayout_grid_back', C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ('adfp_relayout_grids', C.c_int, [C.c_int, C.POINTER(AdfpRelayoutJob), C.c_int, C.c_void_p]),
    ('adfp_relayout_tsdf', C.c_int, [C.POINTER(AdfpTsdf), C.c_void_p, C.c_void_p]),
    ('adfp_pack_decoder', C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_pack_attention', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_decoder_packed_h_words', C.c_longlong, [C.c_int]),
    ('adfp_pack_decoder_h', C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_pack_split_image', C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_pack_images', C.c_int, [C.c_int, C.POINTER(AdfpPackJob), C.c_void_p, C.c_void_p]),
    ('adfp_decoder_packed_ht_words', C.c_longlong, [C.c_int]),
    ('adfp_pack_decoder_ht', C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_train_act_floats', C.c_longlong, [C.c_int]),
    ('adfp_attention_packed_ht_words', C.c_longlong, []),
    ('adfp_pack_attention_ht', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_attention_packed_h_words', C.c_longlong, []),
    ('adfp_pack_attention_h', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_get_rays', C.c_int, [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_rays_from_uv', C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_rays_from_uv_backward', C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_prefilter_rays', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    ('adfp_sample_rays', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Bound),
                                   C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_eval_points', C.c_int, [C.POINTER(AdfpScene), C.POINTER(AdfpPoints), C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ('adfp_eval_points_train', C.c_int, [C.POINTER(AdfpScene), C.POINTER(AdfpPoints), C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(AdfpTrainState), C.c_void_p]),
    ('adfp_eval_points_backward', C.c_int, [C.POINTER(AdfpScene), C.POINTER(AdfpPoints), C.POINTER(AdfpPointsBackwardArgs), C.c_void_p]),
    ('adfp_sample_tsdf', C.c_int, [C.POINTER(AdfpTsdf), C.POINTER(Bound), C.POINTER(AdfpPoints),
                                   C.c_void_p, C.c_void_p]),
    ('adfp_tsdf_integrate', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float * 3),
                                      C.c_float, C.POINTER(C.c_float * 9), C.POINTER(C.c_float * 16), C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    ('adfp_composite', C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_render_forward', C.c_int, [C.POINTER(AdfpScene), C.POINTER(AdfpRenderArgs), C.c_void_p]),
    ('adfp_frustum_mask', C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(Bound), C.POINTER(C.c_float * 16),
                                    C.POINTER(C.c_float * 16), C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_masked_adam', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int,
                                   C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    ('adfp_prefilter_mask', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_mapper_loss', C.c_int, [C.POINTER(AdfpLossArgs), C.c_void_p]),
    ('adfp_mapper_loss_scratch_bytes', C.c_size_t, [C.c_int]),
    ('adfp_mapper_loss_step', C.c_int, [C.POINTER(AdfpLossArgs), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float),
                                        C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    ('adfp_adam_prep', C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    ('adfp_masked_adam_dev', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int,
                                       C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    ('adfp_masked_adam_multi', C.c_int, [C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    ('adfp_adam_grids_cl', C.c_int, [C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    ('adfp_adam_step', C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    ('adfp_camera_from_tensor', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_camera_from_tensor_backward', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_select_pixels', C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_sample_keyframes', C.c_int, [C.c_int, C.POINTER(AdfpKeyframe), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_tracker_head', C.c_int, [C.POINTER(AdfpTrackerHeadArgs), C.c_void_p]),
    ('adfp_tracker_tail', C.c_int, [C.POINTER(AdfpTrackerTailArgs), C.c_void_p]),
    ('adfp_tracker_loss', C.c_int, [C.POINTER(AdfpTrackLossArgs), C.c_void_p]),
    ('adfp_track_keep_best', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_sort_workspace_bytes', C.c_size_t, [C.c_longlong]),
    ('adfp_sort_pairs', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    ('adfp_ray_sort_keys', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Bound), C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_ray_order_probe', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    ('adfp_gather_pack', C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_longlong, C.c_void_p, C.c_void_p]),
    ('adfp_gather_unpack', C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_longlong, C.POINTER(C.c_longlong), C.c_void_p, C.c_void_p]),
    ('adfp_tsdf_stage', C.c_int, [C.POINTER(AdfpScene), C.POINTER(AdfpPoints), C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('adfp_backward_workspace_bytes', C.c_size_t, [C.c_longlong]),
    ('adfp_render_backward', C.c_int, [C.POINTER(AdfpScene), C.POINTER(AdfpBackwardArgs), C.c_void_p]),
    ('adfp_decode_stage', C.c_int, [C.POINTER(AdfpScene), C.POINTER(AdfpPoints), C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    ('adfp_decode_single', C.c_int, [C.POINTER(AdfpScene), C.POINTER(AdfpPoints), C.c_int, C.c_void_p, C.c_void_p]),
    ('adfp_attention_rows', C.c_int, [C.POINTER(AdfpScene), C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
]

_lib = None


def lib():
    """Load libadfp.so once.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: build the HIP extension first '
                '(python -c "import __graft_entry__ as g; g.build()" or attentive_dfprior_amd/csrc/build.sh). '
                'attentive_dfprior_amd has no CPU fallback.')
        handle = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.adfp_version() != ABI_VERSION:
            raise RuntimeError(f'{LIB_PATH} is ABI version {handle.adfp_version()}, this package binds {ABI_VERSION}: rebuild it')
        if HOST_TIMING is not None:
            class _Timed(object):
                pass
            t = _Timed()
            for name, _, _ in SYMBOLS:
                setattr(t, name, timed(getattr(handle, name), name))
            handle = t
        _lib = handle
    return _lib


# ADFP_HOST_TIMING=1: host seconds spent inside each C entry point (name -> [calls, seconds]); tools/host_breakdown.py prints it
HOST_TIMING = {} if os.environ.get('ADFP_HOST_TIMING') else None


def timed(fn, name):
    import time

    def call(*a):
        t0 = time.perf_counter()
        rc = fn(*a)
        e = HOST_TIMING.setdefault(name, [0, 0.0])
        e[0] += 1
        e[1] += time.perf_counter() - t0
        return rc
    return call


def check(rc, what):
    if rc == 0:
        return
    if rc < 0:
        raise RuntimeError(f'{what}: {ERRORS.get(rc, rc)}')
    raise RuntimeError(f'{what}: hipError_t {rc}')


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def current_stream(device):
    """The raw hipStream_t of torch's current stream on `device` (torch.cuda.current_stream(device).cuda_stream costs ~7 us of
    Python object construction per call; a training iteration asks a dozen times)."""
    import torch
    idx = device.index if isinstance(device, torch.device) else torch.device(device).index
    if idx is None:
        idx = torch._C._cuda_getDevice()
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(idx))


class device_guard(object):
    """`with device_guard(dev):` = `with torch.cuda.device(dev):` -- the kernels launch on the CURRENT HIP device -- without the
    per-call argument parsing: switches only when `dev` is not the current device already."""
    __slots__ = ('idx', 'prev')

    def __init__(self, dev):
        self.idx = dev.index if dev.index is not None else -1
        self.prev = -1

    def __enter__(self):
        import torch
        if self.idx >= 0:
            cur = torch._C._cuda_getDevice()
            if cur != self.idx:
                torch._C._cuda_setDevice(self.idx)
                self.prev = cur
        return self

    def __exit__(self, *exc):
        if self.prev >= 0:
            import torch
            torch._C._cuda_setDevice(self.prev)
            self.prev = -1
        return False


def fill_bound(dst, values):
    """values: nested [[lo,hi]]*3 python floats."""
    for k in range(3):
        dst[k][0] = float(values[k][0])
        dst[k][1] = float(values[k][1])


def require_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError(
            f'{name} is on {t.device}: attentive_dfprior_amd runs only on an MI355X through libadfp.so; '
            'there is no CPU fallback.')


# ---- sticky status words (adfp_scene.status) ---------------------------------------------------------------
# A word of pinned host memory: the kernels OR into it with a system-scope atomic (pinned host memory is device-visible under
# HIP's unified addressing), and the host reads it without synchronising.  Every DF module owns one (decoder.DF.status_word), so
# that an f16-range event is attributed to the network object it happened in; sub-networks called on their own share the
# process-wide word below.  An f16-range bit is NOT an error any more: the call that raised it repaired itself on the device
# (adfp_scene.flat_*, csrc/adfp_fallback.h), and the host answers by switching that network to its exact image (read_status ->
# DF._exact_latch).
_status = None


def new_status_word():
    import torch
    return torch.zeros(4, dtype=torch.int32).pin_memory()


def status_word():
    global _status
    if _status is None:
        _status = new_status_word()
    return _status


def status_ptr():
    return C.c_void_p(status_word().data_ptr())


_status_views = {}


def read_status(word, sync=False, device=None):
    """The bits an earlier call left in `word` (cleared on read).  sync=True waits for the device first.  Range bits are returned
    to the caller; any other bit is an error.  The word is read through a ctypes view of the pinned memory: indexing the tensor
    costs two dispatcher calls per render call."""
    if word is None:
        return 0
    if sync:
        import torch
        torch.cuda.synchronize(device)
    addr = word.data_ptr()
    view = _status_views.get(addr)
    if view is None or view[1]() is not word:
        import weakref
        view = _status_views[addr] = (C.c_int.from_address(addr), weakref.ref(word))
    v = view[0].value
    if v:
        view[0].value = 0
        if v & ~STATUS_F16_RANGE:
            raise RuntimeError(f'libadfp: status word {v:#x}')
    return v


def check_status(sync=False, device=None):
    """Process-wide word (sub-networks called on their own): returns its range bits, raises on anything else."""
    return read_status(_status, sync, device)
