"""
Drop-in for the hot-path functions of the reference's ``src/common.py`` (rays, compositing,
coordinate normalisation).  Signatures and argument meaning follow the reference so that
``src/Mapper.py`` / ``src/Tracker.py`` / ``src/utils/Visualizer.py`` call them unchanged.

Ray generation on a GPU device and compositing run in libadfp.so.  Pixel selection keeps
``torch.randint`` as its RNG so the index stream is the reference's (src/common.py:101).
The pose utilities of the Tracker (quad2rotation, get_camera_from_tensor, get_tensor_from_camera, src/common.py:139-203) are
plain torch / numpy here -- host API; inside the fused tracking iteration (tracking.TrackerIteration) the same conversion and its
backward are device kernels (adfp_camera_from_tensor).
"""

import numpy as np
import torch

from . import _lib
from ._lib import lib, ptr, check


def _pixel_dirs(i, j, fx, fy, cx, cy):
    # camera-frame direction of pixel (i, j): x right, y up, looking along -z (unnormalised)
    return torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)


def get_rays(H, W, fx, fy, cx, cy, c2w, device):
    """Rays of a whole image -> (rays_o, rays_d) [H,W,3] (reference src/common.py:254-272)."""
    if isinstance(c2w, np.ndarray):
        c2w = torch.from_numpy(c2w)
    dev = torch.device(device)
    if dev.type == 'cuda':
        m = c2w.detach().to(dev, torch.float32).contiguous()
        with _lib.device_guard(dev):
            ro = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
            rd = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
            check(lib().adfp_get_rays(H, W, fx, fy, cx, cy, ptr(m), ptr(ro), ptr(rd), _lib.current_stream(dev)),
                  'adfp_get_rays')
        return ro, rd
    # host tensors (dataset / keyframe bookkeeping on the CPU): plain indexing arithmetic
    jj, ii = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                            indexing='ij')
    dirs = _pixel_dirs(ii, jj, fx, fy, cx, cy).to(device)
    rot = c2w[:3, :3].to(device)
    rays_d = (dirs[..., None, :] * rot).sum(-1)
    rays_o = c2w[:3, -1].to(device).expand(rays_d.shape)
    return rays_o, rays_d


class _RaysFromUV(torch.autograd.Function):
    """adfp_rays_from_uv / adfp_rays_from_uv_backward: rays through given pixels, differentiable in the camera pose."""

    @staticmethod
    def forward(ctx, c2w, i, j, fx, fy, cx, cy):
        dev = i.device
        n = i.numel()
        pi, pj = i.detach().reshape(-1).float().contiguous(), j.detach().reshape(-1).float().contiguous()
        m = c2w.detach().to(dev, torch.float32).contiguous()
        with _lib.device_guard(dev):
            ro = torch.empty((n, 3), dtype=torch.float32, device=dev)
            rd = torch.empty((n, 3), dtype=torch.float32, device=dev)
            check(lib().adfp_rays_from_uv(ptr(pi), ptr(pj), n, fx, fy, cx, cy, ptr(m), ptr(ro), ptr(rd), _lib.current_stream(dev)),
                  'adfp_rays_from_uv')
        ctx.pix = (pi, pj, fx, fy, cx, cy)
        ctx.c2w_meta = (tuple(c2w.shape), c2w.dtype)
        return ro, rd

    @staticmethod
    def backward(ctx, g_o, g_d):
        pi, pj, fx, fy, cx, cy = ctx.pix
        dev = pi.device
        with _lib.device_guard(dev):
            g = torch.empty((4, 4), dtype=torch.float32, device=dev)
            go = None if g_o is None else g_o.float().contiguous()
            gd = None if g_d is None else g_d.float().contiguous()
            check(lib().adfp_rays_from_uv_backward(ptr(pi), ptr(pj), pi.numel(), fx, fy, cx, cy, ptr(go), ptr(gd), ptr(g),
                                                   _lib.current_stream(dev)), 'adfp_rays_from_uv_backward')
        shape, dtype = ctx.c2w_meta
        g = g[:shape[0], :shape[1]].to(dtype)
        return g, None, None, None, None, None, None


def get_rays_from_uv(i, j, c2w, H, W, fx, fy, cx, cy, device):
    """Rays through the given pixel coordinates (reference src/common.py:76-91).  On a GPU device the rays come from
    libadfp.so and are differentiable in ``c2w`` (camera tracking / bundle adjustment)."""
    if isinstance(c2w, np.ndarray):
        c2w = torch.from_numpy(c2w).to(device)
    if torch.device(device).type == 'cuda' and i.is_cuda:
        return _RaysFromUV.apply(c2w, i, j, float(fx), float(fy), float(cx), float(cy))
    dirs = _pixel_dirs(i, j, fx, fy, cx, cy).to(device).reshape(-1, 1, 3)
    rays_d = (dirs * c2w[:3, :3]).sum(-1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def select_uv(i, j, n, depth, color, device='cuda:0'):
    """n uniformly random pixels out of the dense (i, j) lattice (reference src/common.py:94-109)."""
    i = i.reshape(-1)
    j = j.reshape(-1)
    pick = torch.randint(i.shape[0], (n,), device=device).clamp(0, i.shape[0])
    return i[pick], j[pick], depth.reshape(-1)[pick], color.reshape(-1, 3)[pick]


def get_sample_uv(H0, H1, W0, W1, n, depth, color, device='cuda:0'):
    """Sample n pixels from the window [H0,H1) x [W0,W1) (reference src/common.py:112-124)."""
    depth = depth[H0:H1, W0:W1]
    color = color[H0:H1, W0:W1]
    jj, ii = torch.meshgrid(torch.linspace(H0, H1 - 1, H1 - H0).to(device),
                            torch.linspace(W0, W1 - 1, W1 - W0).to(device), indexing='ij')
    return select_uv(ii, jj, n, depth, color, device=device)


def get_samples(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, c2w, depth, color, device):
    """n random rays of one frame with their depth / colour (reference src/common.py:127-136).

    On the GPU the reference's dozen torch ops (two linspaces, a meshgrid, two window slices, the draw, four gathers) are ONE
    draw -- the same ``torch.randint`` call on the same range, so the index stream is the reference's -- and ONE kernel
    (``adfp_select_pixels``: pixel coordinates, sensor depth and colour of the drawn pixels), followed by the ray kernel."""
    dev = torch.device(device)
    if (dev.type == 'cuda' and isinstance(depth, torch.Tensor) and depth.is_cuda and color.is_cuda
            and depth.dtype == torch.float32 and color.dtype == torch.float32 and depth.dim() == 2 and tuple(color.shape) == (depth.shape[0], depth.shape[1], 3)):
        Hd, Wd = depth.shape
        with _lib.device_guard(depth.device):
            pick = torch.randint((H1 - H0) * (W1 - W0), (n,), device=depth.device).clamp(0, (H1 - H0) * (W1 - W0))     # src/common.py:101-102
            d, c = depth.contiguous(), color.contiguous()
            i = torch.empty((n,), dtype=torch.float32, device=depth.device)
            j = torch.empty_like(i)
            sample_depth = torch.empty_like(i)
            sample_color = torch.empty((n, 3), dtype=torch.float32, device=depth.device)
            check(lib().adfp_select_pixels(ptr(pick), n, H0, H1, W0, W1, Hd, Wd, ptr(d), ptr(c), ptr(i), ptr(j), ptr(sample_depth),
                                           ptr(sample_color), _lib.current_stream(depth.device)), 'adfp_select_pixels')
    else:
        i, j, sample_depth, sample_color = get_sample_uv(H0, H1, W0, W1, n, depth, color, device=device)
    rays_o, rays_d = get_rays_from_uv(i, j, c2w, H, W, fx, fy, cx, cy, device)
    return rays_o, rays_d, sample_depth, sample_color


def get_samples_multi(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, frames, device, out=None):
    """The Mapper's ray batch of one iteration (reference src/Mapper.py:421-436): ``get_samples`` for every ``(c2w, depth, color)`` of
    ``frames`` and the four ``torch.cat`` after it, as the per-frame ``torch.randint`` draws (in order: the index stream of the
    sequential calls) and ONE kernel (``adfp_sample_keyframes``).  Returns ``(rays_o, rays_d, gt_depth, gt_color)`` with ``n`` rows per
    frame, equal bit for bit to the concatenated ``get_samples`` results; ``out`` = four tensors to fill instead (the static input
    buffers of ``mapping.MapperIteration.input_buffers``).  No gradient towards the poses: a pose that requires grad, a frame on the
    host or more than 16 frames take the per-frame path."""
    dev = torch.device(device)
    fast = dev.type == 'cuda' and 0 < len(frames) <= _lib.KEYFRAMES_MAX
    for c2w, depth, color in frames:
        fast = fast and (isinstance(depth, torch.Tensor) and depth.is_cuda and color.is_cuda and depth.dtype == torch.float32
                         and color.dtype == torch.float32 and depth.dim() == 2 and tuple(color.shape) == (depth.shape[0], depth.shape[1], 3)
                         and tuple(depth.shape) == tuple(frames[0][1].shape) and depth.device == frames[0][1].device
                         and not (isinstance(c2w, torch.Tensor) and c2w.requires_grad and torch.is_grad_enabled()))
    if not fast:
        parts = [get_samples(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, c2w, depth, color, device) for c2w, depth, color in frames]
        res = tuple(torch.cat([p[k].float() for p in parts]) for k in range(4))
        if out is not None:
            for dst, src in zip(out, res):
                dst.copy_(src)
            return tuple(out)
        return res
    d0 = frames[0][1]
    Hd, Wd = d0.shape
    total = n * len(frames)
    with _lib.device_guard(d0.device):
        if out is None:
            out = (torch.empty((total, 3), dtype=torch.float32, device=d0.device), torch.empty((total, 3), dtype=torch.float32, device=d0.device),
                   torch.empty((total,), dtype=torch.float32, device=d0.device), torch.empty((total, 3), dtype=torch.float32, device=d0.device))
        for t, shape in zip(out, ((total, 3), (total, 3), (total,), (total, 3))):
            _lib.require_cuda(t, 'out')
            if tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError(f'get_samples_multi: out tensors must be contiguous float32 of shapes [{total},3], [{total},3], [{total}], [{total},3]')
        jobs = (_lib.AdfpKeyframe * len(frames))()
        keep = []
        for k, (c2w, depth, color) in enumerate(frames):
            pick = torch.randint((H1 - H0) * (W1 - W0), (n,), device=d0.device)            # src/common.py:101 (its clamp changes nothing)
            d, c = depth.contiguous(), color.contiguous()
            keep += [pick, d, c]
            jobs[k].idx, jobs[k].depth_img, jobs[k].color_img = pick.data_ptr(), d.data_ptr(), c.data_ptr()
            if isinstance(c2w, torch.Tensor) and c2w.is_cuda:
                m = c2w.detach().float().contiguous()                     # [4,4] or [3,4]: rows 0-2 are read with a stride of 4
                if tuple(m.shape) not in ((4, 4), (3, 4)):
                    raise ValueError(f'c2w: expected [4,4] or [3,4], got {tuple(m.shape)}')
                keep.append(m)
                jobs[k].c2w = m.data_ptr()
            else:
                m = np.asarray(c2w.detach().cpu().numpy() if isinstance(c2w, torch.Tensor) else c2w, dtype=np.float32)
                jobs[k].c2w = None
                jobs[k].c2w_host[:] = [float(v) for v in m[:3, :4].reshape(-1)]
        check(lib().adfp_sample_keyframes(len(frames), jobs, n, H0, H1, W0, W1, Hd, Wd, float(fx), float(fy), float(cx), float(cy),
                                          ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), _lib.current_stream(d0.device)), 'adfp_sample_keyframes')
    return tuple(out)


def filter_rays_in_bound(batch_rays_o, batch_rays_d, batch_gt_depth, batch_gt_color, bound):
    """The Mapper's pre-filter "should pre-filter those out of bounding box depth value"
    (reference src/Mapper.py:438-449): keeps the rays whose sensor depth lies inside the bounding box,
    ``min_axis max_side((bound - o) / d) >= gt_depth``, in their original order.

    Returns ``(rays_o, rays_d, gt_depth, gt_color)`` like the four boolean-mask indexings of the
    reference; the mask and its ordered compaction run in libadfp.so (``adfp_prefilter_rays``), the
    row gather is ``index_select`` so that autograd to camera tensors (BA) is preserved.  Reading the
    kept count synchronises the stream exactly as the reference's boolean indexing does."""
    _lib.require_cuda(batch_rays_o, 'batch_rays_o')
    dev = batch_rays_o.device
    n = batch_rays_o.shape[0]
    with _lib.device_guard(dev):
        ro = batch_rays_o.detach().float().contiguous()
        rd = batch_rays_d.detach().float().contiguous()
        gd = batch_gt_depth.detach().float().contiguous()
        b = torch.as_tensor(bound).to(dev, torch.float64).contiguous()
        idx = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
        cnt = torch.empty((1,), dtype=torch.int32, device=dev)
        check(lib().adfp_prefilter_rays(ptr(ro), ptr(rd), ptr(gd), n, ptr(b), ptr(idx), ptr(cnt),
                                        _lib.current_stream(dev)), 'adfp_prefilter_rays')
        keep = idx[:int(cnt.item())].long()
    return (batch_rays_o.index_select(0, keep), batch_rays_d.index_select(0, keep),
            batch_gt_depth.index_select(0, keep), batch_gt_color.index_select(0, keep))


def quad2rotation(quad):
    """Batch of quaternions (r, i, j, k), not necessarily unit -> rotation matrices [B,3,3], differentiable
    (reference src/common.py:139-163)."""
    qr, qi, qj, qk = quad[:, 0], quad[:, 1], quad[:, 2], quad[:, 3]
    two_s = 2.0 / (quad * quad).sum(-1)
    rows = [1 - two_s * (qj ** 2 + qk ** 2), two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr),
            two_s * (qi * qj + qk * qr), 1 - two_s * (qi ** 2 + qk ** 2), two_s * (qj * qk - qi * qr),
            two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), 1 - two_s * (qi ** 2 + qj ** 2)]
    return torch.stack(rows, -1).reshape(-1, 3, 3)


class _CameraFromTensor(torch.autograd.Function):
    """adfp_camera_from_tensor / _backward: ONE camera tensor on the GPU -> [3,4] camera-to-world in one launch each way.  The
    reference's quad2rotation is ~40 small torch ops forward and as many autograd nodes backward (src/common.py:139-178) -- a
    quarter of a 200-ray tracking iteration's host time; same float32 arithmetic, operation by operation (mini_pose.npz)."""

    @staticmethod
    def forward(ctx, cam):
        dev = cam.device
        t = cam.detach().contiguous()
        with _lib.device_guard(dev):
            m = torch.empty((4, 4), dtype=torch.float32, device=dev)
            check(lib().adfp_camera_from_tensor(ptr(t), ptr(m), _lib.current_stream(dev)), 'adfp_camera_from_tensor')
        ctx.cam = t
        return m[:3]

    @staticmethod
    def backward(ctx, g):
        t = ctx.cam
        dev = t.device
        with _lib.device_guard(dev):
            g12 = g.float().contiguous()          # the kernel reads rows 0-2 of a row-major [*,4] cotangent: 12 floats
            out = torch.empty((7,), dtype=torch.float32, device=dev)
            check(lib().adfp_camera_from_tensor_backward(ptr(t), ptr(g12), ptr(out), _lib.current_stream(dev)),
                  'adfp_camera_from_tensor_backward')
        return out


def get_camera_from_tensor(inputs):
    """Quaternion + translation [7] or [B,7] -> [3,4] / [B,3,4] camera-to-world (reference src/common.py:166-178)."""
    if inputs.dim() == 1 and inputs.is_cuda and inputs.dtype == torch.float32 and inputs.shape[0] == 7:
        return _CameraFromTensor.apply(inputs)
    single = inputs.dim() == 1
    if single:
        inputs = inputs.unsqueeze(0)
    RT = torch.cat([quad2rotation(inputs[:, :4]), inputs[:, 4:, None]], 2)
    return RT[0] if single else RT


def get_tensor_from_camera(RT, Tquad=False):
    """Camera-to-world matrix -> float32 [7] quaternion (r, i, j, k; r >= 0 branch by the largest diagonal term) + translation, on
    the matrix's device (reference src/common.py:181-203, which goes through mathutils' Matrix.to_quaternion; the same
    rotation either way, q and -q being one rotation)."""
    dev = RT.device if isinstance(RT, torch.Tensor) else None
    M = RT.detach().cpu().numpy() if isinstance(RT, torch.Tensor) else np.asarray(RT)
    R, T = M[:3, :3].astype(np.float64), M[:3, 3].astype(np.float64)
    # mathutils' Matrix.to_quaternion normalises the matrix first (unit-length axis vectors; a left-handed one is negated) and
    # returns a unit quaternion: the constant-speed guess delta @ pre_c2w (src/Tracker.py:213-215) is only orthonormal up to
    # accumulated float error, and the raw quaternion components are what Adam steps on and the error log prints
    norms = np.linalg.norm(R, axis=0)
    R = R / np.where(norms > 0, norms, 1.0)
    if np.linalg.det(R) < 0:
        R = -R
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = 2.0 * np.sqrt(1.0 + tr)
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        k = int(np.argmax(np.diag(R)))
        a, b = (k + 1) % 3, (k + 2) % 3
        s = 2.0 * np.sqrt(1.0 + R[k, k] - R[a, a] - R[b, b])
        v = [0.0, 0.0, 0.0]
        v[k], v[a], v[b] = 0.25 * s, (R[a, k] + R[k, a]) / s, (R[b, k] + R[k, b]) / s
        q = [(R[b, a] - R[a, b]) / s] + v
    q = np.asarray(q)
    q = q / np.linalg.norm(q)
    if q[0] < 0:
        q = -q
    out = torch.from_numpy(np.concatenate([T, q] if Tquad else [q, T])).float()
    return out.to(dev) if dev is not None and dev.type != 'cpu' else out


def random_select(l, k):
    """k distinct indices out of range(l) (reference src/common.py:68-73)."""
    return list(np.random.permutation(np.arange(l))[:min(l, k)])


def normalize_3d_coordinate(p, bound):
    """Map world coordinates into [-1, 1] of ``bound`` IN PLACE, like the reference
    (src/common.py:275-290); the kernels do this internally, the function is kept for callers."""
    p = p.reshape(-1, 3)
    for k in range(3):
        p[:, k] = ((p[:, k] - bound[k, 0]) / (bound[k, 1] - bound[k, 0])) * 2 - 1.0
    return p


def raw2outputs_nerf_color(raw, z_vals, rays_d, occupancy=False, device='cuda:0'):
    """Alpha compositing (reference src/common.py:206-251, ``occupancy=True`` branch) on the GPU.

    raw [N,S,4] (rgb, occ), z_vals [N,S] -> depth [N], depth_var [N], rgb [N,3], weights [N,S].
    depth / variance are float64 like the reference's when z_vals is float64.
    """
    if not occupancy:
        raise NotImplementedError('libadfp implements the occupancy=True branch (configs/df_prior.yaml:4)')
    _lib.require_cuda(raw, 'raw')
    dev = raw.device
    N, S = raw.shape[0], raw.shape[1]
    with _lib.device_guard(dev):
        r = raw.detach().float().contiguous()
        z = z_vals.detach().to(torch.float64).contiguous()
        depth = torch.empty((N,), dtype=torch.float64, device=dev)
        var = torch.empty((N,), dtype=torch.float64, device=dev)
        rgb = torch.empty((N, 3), dtype=torch.float32, device=dev)
        wts = torch.empty((N, S), dtype=torch.float32, device=dev)
        check(lib().adfp_composite(ptr(r), ptr(z), N, S, ptr(depth), ptr(var), ptr(rgb), ptr(wts),
                                   _lib.current_stream(dev)), 'adfp_composite')
    if z_vals.dtype != torch.float64:
        depth, var = depth.to(z_vals.dtype), var.to(z_vals.dtype)
    return depth, var, rgb, wts


def sample_pdf(*args, **kwargs):
    """Hierarchical resampling is dead code in the reference (``N_importance: 0``,
    configs/df_prior.yaml:96; its enabled branch re-evaluates stale points, Renderer.py:246)."""
    raise NotImplementedError('N_importance > 0 is not supported (dead in the reference)')
