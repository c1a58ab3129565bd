"""Autograd bridge for Renderer.render_batch_ray: the training path of src/Mapper.py:451-473.

Forward = adfp_render_forward with a caller-owned training state; backward = adfp_render_backward
(include/adfp.h).  Gradients are produced for the three feature grids (dense, in the shape of
``c[key]``, which may be an autograd non-leaf built by index_put, src/Mapper.py:382-388) and for every
decoder parameter that requires grad, and for the rays (``rays_o`` / ``rays_d``: the camera pose of
src/Tracker.py:112-133 reaches the renderer only through them).
"""
import torch

_GRIDS = (('low', 'grid_low'), ('high', 'grid_high'), ('color', 'grid_color'))
_NETS = (('low', 'low_decoder'), ('high', 'high_decoder'), ('color', 'color_decoder'), ('att', 'mlp'))


_USED = {'low': ('low',), 'high': ('low', 'high', 'att'), 'color': ('low', 'high', 'color', 'att')}


def _trainable(decoders, stage):
    """The parameters autograd has to see: those of the stage's networks that require grad -> (tensors, [(net, indices)])."""
    tensors, slots = [], []
    for name in _USED[stage]:
        idx = [k for k, p in enumerate(decoders.net_params(name)) if p.requires_grad]
        if idx:
            params = decoders.net_params(name)
            slots.append((name, idx))
            tensors.extend([params[k] for k in idx])
    return tensors, slots


def _param_grads(decoders, slots, flats, needs, off):
    """Per-parameter gradient views of the flat gradients, in the order _trainable handed the parameters out."""
    out = []
    for name, idx in slots:
        flat = flats.get(name)
        if flat is None:
            out.extend([None] * len(idx))
            off += len(idx)
            continue
        params = decoders.net_params(name)
        pieces = flat.split_with_sizes([p.numel() for p in params])          # views of the flat gradient
        for k in idx:
            if needs[off]:
                p, g = params[k], pieces[k]
                if p.dim() != 1:
                    g = g.view(p.shape)
                out.append(g if g.dtype == p.dtype else g.to(p.dtype))
            else:
                out.append(None)
            off += 1
    return out


class _RenderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bundle, rays_o, rays_d, grid_low, grid_high, grid_color, *params):
        (engine, decoders, gt_depth, tsdf_volume, tsdf_bnds, bound, stage, n_samples, n_surface,
         lindisp, perturb, t_rand, depth_max, slots) = bundle
        c = {'grid_low': grid_low, 'grid_high': grid_high, 'grid_color': grid_color}
        need_flat = {name: False for name in ('low', 'high', 'color', 'att')}
        for name, _ in slots:
            need_flat[name] = True
        depth, unc, color, weight, saved = engine.render_forward(
            decoders, c, rays_o, rays_d, gt_depth, tsdf_volume, tsdf_bnds, bound, stage, n_samples, n_surface,
            lindisp, perturb, t_rand, depth_max, train=True, need_flat=need_flat)
        ctx.bundle = bundle
        ctx.saved = saved
        ctx.c = c
        ctx.set_materialize_grads(False)
        return depth, unc, color, weight

    @staticmethod
    def backward(ctx, g_depth, g_unc, g_color, g_weight):
        (engine, decoders, gt_depth, tsdf_volume, tsdf_bnds, bound, stage, *_rest, slots) = ctx.bundle
        used = _USED[stage]
        needs = ctx.needs_input_grad
        need_rays = bool(needs[1] or needs[2])
        need_grid = {name: bool(needs[3 + k]) and name in used for k, (name, _) in enumerate(_GRIDS)}
        need_flat, off = {}, 6
        for name, idx in slots:
            need_flat[name] = any(needs[off:off + len(idx)])
            off += len(idx)
        if g_weight is not None:
            g_weight = g_weight.reshape(g_weight.shape[0], -1)
        if ctx.saved is None:                                 # zero rays (an empty shard): zero gradients in the usual bucket layout
            dev = ctx.c['grid_low'].device
            grids, flats = engine.zero_grad_bucket(ctx.c, need_grid, need_flat, dev)
            g_rays = (torch.zeros((0, 3), device=dev), torch.zeros((0, 3), device=dev))
        else:
            grids, flats, g_rays = engine.render_backward(decoders, ctx.c, tsdf_volume, tsdf_bnds, bound, stage, ctx.saved,
                                                          g_depth, g_unc, g_color, g_weight, need_grid, need_flat, need_rays)
        out = [None, g_rays[0] if needs[1] else None, g_rays[1] if needs[2] else None]
        for k, (name, key) in enumerate(_GRIDS):
            g = grids.get(name)
            if g is not None and ctx.c[key].dtype != g.dtype:
                g = g.to(ctx.c[key].dtype)
            out.append(g)
        out.extend(_param_grads(decoders, slots, flats, needs, 6))
        ctx.saved = None
        return tuple(out)


def render_with_grad(engine, decoders, c, rays_o, rays_d, gt_depth, tsdf_volume, tsdf_bnds, bound, stage, n_samples,
                     n_surface, lindisp, perturb, t_rand, depth_max, need_param_grad=True):
    if n_samples + (n_surface if gt_depth is not None else 0) > 256:
        raise NotImplementedError('training path supports at most 256 samples per ray')
    params, slots = _trainable(decoders, stage) if need_param_grad else ([], [])
    bundle = (engine, decoders, gt_depth, tsdf_volume, tsdf_bnds, bound, stage, n_samples, n_surface,
              lindisp, perturb, t_rand, depth_max, slots)
    return _RenderFn.apply(bundle, rays_o, rays_d, c['grid_low'], c['grid_high'], c['grid_color'], *params)


class _EvalPointsFn(torch.autograd.Function):
    """Renderer.eval_points / DF.forward under autograd (the reference's are plain torch ops, src/utils/Renderer.py:27-71,
    src/conv_onet/models/decoder.py:307-353): gradients for the query points, the feature grids and the decoder parameters."""

    @staticmethod
    def forward(ctx, bundle, pts, grid_low, grid_high, grid_color, *params):
        engine, decoders, tsdf_volume, tsdf_bnds, bound, stage, apply_bound, slots = bundle
        c = {'grid_low': grid_low, 'grid_high': grid_high, 'grid_color': grid_color}
        need_flat = {name: False for name in ('low', 'high', 'color', 'att')}
        for name, _ in slots:
            need_flat[name] = True
        raw, w, saved = engine.eval_points_forward(decoders, pts, c, tsdf_volume, tsdf_bnds, bound, stage, apply_bound, train=True,
                                                   need_flat=need_flat)
        ctx.bundle, ctx.saved, ctx.c = bundle, saved, c
        ctx.pts_dtype = pts.dtype
        ctx.set_materialize_grads(False)
        return raw, w

    @staticmethod
    def backward(ctx, g_raw, g_w):
        engine, decoders, tsdf_volume, tsdf_bnds, bound, stage, apply_bound, slots = ctx.bundle
        used = _USED[stage]
        needs = ctx.needs_input_grad
        need_grid = {name: bool(needs[2 + k]) and name in used for k, (name, _) in enumerate(_GRIDS)}
        need_flat, off = {}, 5
        for name, idx in slots:
            need_flat[name] = any(needs[off:off + len(idx)])
            off += len(idx)
        if ctx.saved is None:                                 # zero points
            return (None,) * len(needs)
        grids, flats, g_pts = engine.eval_points_backward(decoders, ctx.c, tsdf_volume, tsdf_bnds, bound, stage, apply_bound, ctx.saved,
                                                          g_raw, g_w, need_grid, need_flat, bool(needs[1]))
        out = [None, None if g_pts is None else g_pts.to(ctx.pts_dtype)]
        for k, (name, key) in enumerate(_GRIDS):
            g = grids.get(name)
            if g is not None and ctx.c[key].dtype != g.dtype:
                g = g.to(ctx.c[key].dtype)
            out.append(g)
        out.extend(_param_grads(decoders, slots, flats, needs, 5))
        ctx.saved = None
        return tuple(out)


def eval_points_with_grad(engine, decoders, pts, c, tsdf_volume, tsdf_bnds, bound, stage, apply_bound):
    params, slots = _trainable(decoders, stage)
    bundle = (engine, decoders, tsdf_volume, tsdf_bnds, bound, stage, apply_bound, slots)
    return _EvalPointsFn.apply(bundle, pts, c['grid_low'], c['grid_high'], c['grid_color'], *params)
