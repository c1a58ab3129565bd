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


class _RenderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bundle, rays_o, rays_d, grid_low, grid_high, grid_color, *params):
        (engine, decoders, gt_depth, tsdf_volume, tsdf_bnds, bound, stage, n_samples, n_surface,
         lindisp, perturb, t_rand, depth_max, need_param_grad) = bundle
        c = {'grid_low': grid_low, 'grid_high': grid_high, 'grid_color': grid_color}
        depth, unc, color, weight, saved = engine.render_forward(
            decoders, c, rays_o, rays_d, gt_depth, tsdf_volume, tsdf_bnds, bound, stage, n_samples, n_surface,
            lindisp, perturb, t_rand, depth_max, train=True, need_flat=None if need_param_grad else {})
        ctx.bundle = bundle
        ctx.saved = saved
        ctx.c = c
        ctx.n_params = [len(decoders.net_params(name)) for name, _ in _NETS]
        ctx.set_materialize_grads(False)
        return depth, unc, color, weight

    @staticmethod
    def backward(ctx, g_depth, g_unc, g_color, g_weight):
        (engine, decoders, gt_depth, tsdf_volume, tsdf_bnds, bound, stage, *_rest) = ctx.bundle
        used = {'low': ('low',), 'high': ('low', 'high', 'att'), 'color': ('low', 'high', 'color', 'att')}[stage]
        need_rays = bool(ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        need_grid = {name: bool(ctx.needs_input_grad[3 + k]) and name in used for k, (name, _) in enumerate(_GRIDS)}
        need_flat, off = {}, 6
        for (name, attr), n in zip(_NETS, ctx.n_params):
            need_flat[name] = ctx.bundle[-1] and name in used and any(ctx.needs_input_grad[off:off + n])
            off += n
        if g_weight is not None:
            g_weight = g_weight.reshape(g_weight.shape[0], -1)
        grids, flats, g_rays = engine.render_backward(decoders, ctx.c, tsdf_volume, tsdf_bnds, bound, stage, ctx.saved,
                                                      g_depth, g_unc, g_color, g_weight, need_grid, need_flat, need_rays)
        out = [None, g_rays[0] if ctx.needs_input_grad[1] else None, g_rays[1] if ctx.needs_input_grad[2] else None]
        for k, (name, key) in enumerate(_GRIDS):
            g = grids.get(name)
            if g is not None and ctx.c[key].dtype != g.dtype:
                g = g.to(ctx.c[key].dtype)
            out.append(g)
        off = 6
        for (name, attr), n in zip(_NETS, ctx.n_params):
            flat = flats.get(name)
            params = decoders.net_params(name)
            if flat is None:
                out.extend([None] * n)
            else:
                pieces = torch.split(flat, [p.numel() for p in params])       # views of the flat gradient
                for p, piece in zip(params, pieces):
                    if ctx.needs_input_grad[off]:
                        g = piece.view(p.shape)
                        out.append(g if g.dtype == p.dtype else g.to(p.dtype))
                    else:
                        out.append(None)
                    off += 1
                continue
            off += n
        ctx.saved = None
        return tuple(out)


def render_with_grad(engine, decoders, c, rays_o, rays_d, gt_depth, tsdf_volume, tsdf_bnds, bound, stage, n_samples,
                     n_surface, lindisp, perturb, t_rand, depth_max, need_param_grad=True):
    if n_samples + (n_surface if gt_depth is not None else 0) > 256:
        raise NotImplementedError('training path supports at most 256 samples per ray')
    bundle = (engine, decoders, gt_depth, tsdf_volume, tsdf_bnds, bound, stage, n_samples, n_surface,
              lindisp, perturb, t_rand, depth_max, need_param_grad)
    params = []
    for name, _ in _NETS:
        params += decoders.net_params(name)
    return _RenderFn.apply(bundle, rays_o, rays_d, c['grid_low'], c['grid_high'], c['grid_color'], *params)


class _EvalPointsFn(torch.autograd.Function):
    """Renderer.eval_points / DF.forward under autograd (the reference's are plain torch ops, src/utils/Renderer.py:27-71,
    src/conv_onet/models/decoder.py:307-353): gradients for the query points, the feature grids and the decoder parameters."""

    @staticmethod
    def forward(ctx, bundle, pts, grid_low, grid_high, grid_color, *params):
        engine, decoders, tsdf_volume, tsdf_bnds, bound, stage, apply_bound = bundle
        c = {'grid_low': grid_low, 'grid_high': grid_high, 'grid_color': grid_color}
        raw, w, saved = engine.eval_points_forward(decoders, pts, c, tsdf_volume, tsdf_bnds, bound, stage, apply_bound, train=True)
        ctx.bundle, ctx.saved, ctx.c = bundle, saved, c
        ctx.pts_dtype = pts.dtype
        ctx.n_params = [len(decoders.net_params(name)) for name, _ in _NETS]
        ctx.set_materialize_grads(False)
        return raw, w

    @staticmethod
    def backward(ctx, g_raw, g_w):
        engine, decoders, tsdf_volume, tsdf_bnds, bound, stage, apply_bound = ctx.bundle
        used = {'low': ('low',), 'high': ('low', 'high', 'att'), 'color': ('low', 'high', 'color', 'att')}[stage]
        need_grid = {name: bool(ctx.needs_input_grad[2 + k]) and name in used for k, (name, _) in enumerate(_GRIDS)}
        need_flat, off = {}, 5
        for (name, attr), n in zip(_NETS, ctx.n_params):
            need_flat[name] = name in used and any(ctx.needs_input_grad[off:off + n])
            off += n
        if ctx.saved is None:                                 # zero points
            return (None,) * (5 + sum(ctx.n_params))
        grids, flats, g_pts = engine.eval_points_backward(decoders, ctx.c, tsdf_volume, tsdf_bnds, bound, stage, apply_bound, ctx.saved,
                                                          g_raw, g_w, need_grid, need_flat, bool(ctx.needs_input_grad[1]))
        out = [None, None if g_pts is None else g_pts.to(ctx.pts_dtype)]
        for k, (name, key) in enumerate(_GRIDS):
            g = grids.get(name)
            if g is not None and ctx.c[key].dtype != g.dtype:
                g = g.to(ctx.c[key].dtype)
            out.append(g)
        off = 5
        for (name, attr), n in zip(_NETS, ctx.n_params):
            flat = flats.get(name)
            params = decoders.net_params(name)
            if flat is None:
                out.extend([None] * n)
            else:
                for p, piece in zip(params, torch.split(flat, [p.numel() for p in params])):
                    out.append(piece.view(p.shape).to(p.dtype) if ctx.needs_input_grad[off] else None)
                    off += 1
                continue
            off += n
        ctx.saved = None
        return tuple(out)


def eval_points_with_grad(engine, decoders, pts, c, tsdf_volume, tsdf_bnds, bound, stage, apply_bound):
    bundle = (engine, decoders, tsdf_volume, tsdf_bnds, bound, stage, apply_bound)
    params = []
    for name, _ in _NETS:
        params += decoders.net_params(name)
    return _EvalPointsFn.apply(bundle, pts, c['grid_low'], c['grid_high'], c['grid_color'], *params)
