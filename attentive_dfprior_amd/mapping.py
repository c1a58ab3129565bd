"""
The Mapper's per-call bookkeeping around the render path, on the MI355X:

``frustum_mask``     = ``Mapper.get_mask_from_c2w`` (reference src/Mapper.py:90-158), which the reference
                       runs on the host with numpy + ``cv2.remap`` once per grid per mapping call;
``MaskedGridAdam``   = the optimisation of the masked part of each feature grid.  The reference keeps a
                       compact copy ``val_grad = val[mask]`` as the Adam parameter and index_puts it into the
                       dense grid before AND after every iteration (src/Mapper.py:347-361, :382-388,
                       :476-482); here the dense grid itself is updated in place on the masked voxels by one
                       kernel per grid -- same values, no boolean-index gathers / scatters, no host sync.

How the reference's ``optimize_map`` would use it (the grids stay ordinary leaf tensors)::

    opt_grids = MaskedGridAdam(c, {k: frustum_mask(cur_c2w, c[k].shape[2:], cur_gt_depth, self.bound,
                                                   H, W, fx, fy, cx, cy) for k in c})
    for joint_iter in range(num_joint_iters):
        ...                                   # render_batch_ray(c, ...) -> loss -> loss.backward()
        opt_grids.step({'grid_low': low_lr, 'grid_high': high_lr, 'grid_color': color_lr})
        optimizer.step()                      # torch Adam over the decoder / mlp parameters only

There is no CPU fallback: everything raises on host tensors.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, ptr, check


def frustum_mask(c2w, val_shape, depth, bound, H, W, fx, fy, cx, cy):
    """Boolean mask ``[Z, Y, X]`` of the grid points selected for optimisation (src/Mapper.py:90-158; the
    reference returns it as ``[X, Y, Z]`` and permutes at the call site, :345).

    c2w [4,4] camera pose, ``val_shape`` = ``val.shape[2:]`` = (Z, Y, X), ``depth`` [H,W] float image on
    the GPU, ``bound`` [3,2]."""
    _lib.require_cuda(depth, 'depth')
    dev = depth.device
    Z, Y, X = (int(v) for v in val_shape)
    m = torch.as_tensor(c2w).detach().to('cpu', torch.float32)
    w2c = torch.linalg.inv(m)                                       # np.linalg.inv(c2w), src/Mapper.py:113
    a_c2w = (C.c_float * 16)(*m.reshape(-1).tolist())
    a_w2c = (C.c_float * 16)(*w2c.reshape(-1).tolist())
    b = _lib.Bound()
    bb = torch.as_tensor(bound).detach().to('cpu', torch.float64)
    for k in range(3):
        b[k][0], b[k][1] = float(bb[k, 0]), float(bb[k, 1])
    with torch.cuda.device(dev):
        d = depth.detach().to(torch.float32).contiguous()
        sampled = torch.empty((X * Y * Z,), dtype=torch.float32, device=dev)
        scratch = torch.empty((1,), dtype=torch.int32, device=dev)
        mask = torch.empty((Z, Y, X), dtype=torch.uint8, device=dev)
        check(lib().adfp_frustum_mask(X, Y, Z, C.byref(b), C.byref(a_c2w), C.byref(a_w2c), float(fx), float(fy), float(cx),
                                      float(cy), int(H), int(W), ptr(d), ptr(sampled), ptr(scratch), ptr(mask),
                                      _lib.current_stream(dev)), 'adfp_frustum_mask')
    return mask.bool()


class MaskedGridAdam:
    """Adam over the masked voxels of the feature grids, in place (see the module docstring).

    ``grids``  dict name -> leaf tensor ``[1, C, Z, Y, X]`` float32 on the GPU (``.grad`` is read by ``step``)
    ``masks``  dict name -> bool/uint8 ``[Z, Y, X]`` (or None = the whole grid, ``frustum_feature_selection: False``)
    State (exp_avg, exp_avg_sq, step count) starts at zero like the fresh optimizer the reference builds in
    every ``optimize_map`` call (src/Mapper.py:374)."""

    def __init__(self, grids, masks=None, betas=(0.9, 0.999), eps=1e-8):
        self.grids = dict(grids)
        self.betas, self.eps = betas, eps
        self.masks, self.state = {}, {}
        for k, g in self.grids.items():
            _lib.require_cuda(g, k)
            if g.dtype != torch.float32 or not g.is_contiguous():
                raise ValueError(f'{k}: expected a contiguous float32 grid')
            mk = None if masks is None else masks.get(k)
            if mk is not None:
                if tuple(mk.shape) != tuple(g.shape[2:]):
                    raise ValueError(f'{k}: mask shape {tuple(mk.shape)} != grid {tuple(g.shape[2:])}')
                mk = mk.to(g.device, torch.uint8).contiguous()
            self.masks[k] = mk
            self.state[k] = [torch.zeros_like(g), torch.zeros_like(g), 0]

    def zero_grad(self):
        for g in self.grids.values():
            g.grad = None

    @torch.no_grad()
    def step(self, lrs):
        """One Adam step per grid with the given learning rates (a grid whose lr is 0 still advances its
        moments, like a torch param group with lr 0); grids without a gradient are skipped like torch does."""
        L = lib()
        for k, g in self.grids.items():
            if g.grad is None:
                continue
            grad = g.grad.contiguous()
            st = self.state[k]
            st[2] += 1
            nvox = g.shape[2] * g.shape[3] * g.shape[4]
            with torch.cuda.device(g.device):
                check(L.adfp_masked_adam(ptr(g), ptr(grad), ptr(st[0]), ptr(st[1]),
                                         ptr(self.masks[k]) if self.masks[k] is not None else None, nvox, g.shape[1],
                                         float(lrs[k]), self.betas[0], self.betas[1], self.eps, st[2],
                                         _lib.current_stream(g.device)), 'adfp_masked_adam')
            torch.autograd.graph.increment_version(g)     # updated through a raw pointer: invalidate the layout caches
