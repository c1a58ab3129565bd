"""GPU (MI355X): backward of render_batch_ray (the Mapper's training step, src/Mapper.py:451-473)
through the HIP kernels against the gradients the REFERENCE's autograd produced
(tests/golden/mini_<stage>.npz, keys g.* / gw.*) and against the oracle's autograd."""
import numpy as np
import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import synthetic
from oracle import adfp_oracle as O
from conftest import (make_cfg, to_dev, assert_close_scale, assert_param_grad_close, assert_grad_tight, ReluCapture,
                      assert_forced_decisions_are_boundary_units)

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
GTOL = 2e-4          # gradients: relative to the largest entry of each tensor


def mapper_loss(depth, color, weight, gt_depth, gt_color, stage, warm):
    m = gt_depth > 0
    loss = torch.abs(gt_depth[m] - depth[m]).sum()
    if warm:
        loss = loss + torch.abs(weight - torch.ones(weight.shape, device=weight.device)).sum()
    if stage == 'color':
        loss = loss + 0.2 * torch.abs(gt_color - color).sum()
    return loss


MODES = ['f32', 'f16x3']      # ADFP_MATH: exact f32-input MFMA forward + backward / the default f16-split forward + backward


def grad_close(got, ref, what, tol=GTOL, mode=None):
    """Grid gradients: 2e-4 of the tensor's scale, ReLU-boundary samples aside (such a sample reaches 8 voxels: 2e-3 of a
    grid's elements is generous).  Parameter gradients: conftest.assert_param_grad_close, per math mode."""
    last = what.split()[-1]
    is_param = any(t in last for t in ('decoder', 'mlp', 'weight', 'bias', '_B')) or last in ('W2',)
    if is_param:
        assert_param_grad_close(got, ref, what, mode)
    else:
        assert_close_scale(got.detach(), ref, tol, what, flip_frac=2e-3 if last.startswith('grid') else 0.0)


def run(mini, stage, warm, sd=None, n_samples=None, n_surface=None, rays=None, bwd_options=None, capture=None, side_lane=None):
    """capture: a dict; the backward then exports its ReLU decisions and capture['relu_masks'] = Engine.relu_masks(...)."""
    sd = mini.sd if sd is None else sd
    dec = A.DF()
    dec.load_state_dict(sd)
    dec.bound = mini.bound
    dec = dec.to(DEV)
    for p in dec.parameters():
        p.requires_grad_(True)
    rend = A.Renderer(make_cfg(n_samples or mini.n_samples, n_surface if n_surface is not None else mini.n_surface),
                      None, mini)
    if bwd_options is not None:
        rend._engine.bwd_options = bwd_options
    if side_lane is not None:                                # the autograd path takes the lane only when told to (default: MapperIteration alone)
        eng = rend._engine
        orig_bwd = eng.render_backward
        eng.render_backward = lambda *a, **k: orig_bwd(*a, **dict(k, side_lane=side_lane))
    if capture is not None:
        cap = ReluCapture(rend)
    c = {k: v.to(DEV).clone().requires_grad_(True) for k, v in mini.c.items()}
    ro, rd, gd, gc = rays if rays is not None else (mini.rays_o, mini.rays_d, mini.gt_depth, mini.gt_color)
    ro, rd, gd, gc = ro.to(DEV), rd.to(DEV), gd.to(DEV), gc.to(DEV)
    d, u, col, w = rend.render_batch_ray(c, dec, rd, ro, DEV, mini.tsdf_volume.to(DEV), mini.tsdf_bnds.to(DEV), stage,
                                         gt_depth=gd)
    loss = mapper_loss(d, col, w, gd, gc, stage, warm)
    loss.backward()
    if capture is not None:
        capture['relu_masks'] = cap.masks(stage)
    return loss, c, dec


def oracle_grads(mini, sd, rays, stage, warm, n_samples, n_surface, relu_masks=None):
    """The oracle's autograd under the Mapper loss -> (loss, {grid: grad}, {parameter: grad}); relu_masks: the kernels' ReLU
    decisions, forced (oracle.adfp_oracle._relu)."""
    c_or = {k: v.clone().requires_grad_(True) for k, v in mini.c.items()}
    sd_or = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ro, rd, gd, gc = rays
    d2, u2, col2, w2 = O.render_batch_ray(sd_or, c_or, rd, ro, mini.tsdf_volume, mini.tsdf_bnds, mini.bound, stage, gd, n_samples,
                                          n_surface, relu_masks=relu_masks)
    loss2 = O.mapper_loss(d2, col2, w2, gd, gc, stage, warm)
    loss2.backward()
    zero = lambda t: t.grad if t.grad is not None else torch.zeros_like(t)           # noqa: E731
    return loss2, {k: zero(v) for k, v in c_or.items()}, {k: zero(v) for k, v in sd_or.items()}


def check_tight(mini, stage, warm, mode, sd=None, n_samples=None, n_surface=None, rays=None, what=''):
    """THE gradient criterion: kernels vs the oracle's autograd along the kernels' own ReLU decisions, every element of every
    grid and parameter gradient within conftest.TIGHT_GRAD_TOL x the tensor's scale; and the forced decisions are legitimate --
    they differ from relu's own only on units whose pre-activation is within rounding of zero."""
    sd = mini.sd if sd is None else sd
    ns = n_samples or mini.n_samples
    nf = n_surface if n_surface is not None else mini.n_surface
    rays = rays if rays is not None else (mini.rays_o, mini.rays_d, mini.gt_depth, mini.gt_color)
    cap = {}
    loss, c, dec = run(mini, stage, warm, sd=sd, n_samples=ns, n_surface=nf, rays=rays, capture=cap)
    O.reset_relu_flips()
    loss2, gc_or, gsd_or = oracle_grads(mini, sd, rays, stage, warm, ns, nf, relu_masks=cap['relu_masks'])
    flips = dict(O.RELU_FLIPS)
    assert_forced_decisions_are_boundary_units(flips)
    assert abs(loss.item() - loss2.item()) <= 1e-5 * abs(loss2.item())
    for k in c:
        got = c[k].grad if c[k].grad is not None else torch.zeros_like(c[k])
        assert_grad_tight(got, gc_or[k], f'{what}{stage} d/d {k}', mode)
    for name, p in dec.named_parameters():
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        assert_grad_tight(got, gsd_or[name], f'{what}{stage} d/d {name}', mode)
    return flips


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('stage', O.STAGES)
@pytest.mark.parametrize('warm', [False, True])
def test_mapper_gradients_along_the_kernels_relu_decisions(mini, stage, warm, mode, monkeypatch):
    """The golden scene (the inputs of tests/golden/mini_<stage>.npz), all three stages, both loss variants, both math modes:
    conftest.TIGHT_GRAD_TOL (5e-5, half the north-star tolerance) of each tensor's scale on EVERY element."""
    monkeypatch.setenv('ADFP_MATH', mode)
    check_tight(mini, stage, warm, mode, what='golden scene, ')


@pytest.mark.parametrize('mode', MODES)
def test_gradients_second_seed_64_samples_along_the_kernels_relu_decisions(mini, mode, monkeypatch):
    """S = 64 (benchmark sampling), other weights, 300 rays of 3 poses (19 200 samples): the case whose unforced comparison shows
    the largest ReLU-boundary deviations (profiles/r03_grad_stats.txt: 5.8e-4 x scale)."""
    monkeypatch.setenv('ADFP_MATH', mode)
    rays = synthetic.make_ray_batch(synthetic.mini_scene(), 300, seed=8, poses=3)
    check_tight(mini, 'color', True, mode, sd=O.random_state_dict(seed=17), n_samples=48, n_surface=16, rays=rays, what='second seed, ')


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('stage', O.STAGES)
@pytest.mark.parametrize('tag,warm', [('g', False), ('gw', True)])
def test_mapper_gradients_vs_reference_golden(mini, stage, tag, warm, mode, monkeypatch):
    monkeypatch.setenv('ADFP_MATH', mode)
    g = mini.golden(stage)
    loss, c, dec = run(mini, stage, warm)
    assert abs(loss.item() - float(g[tag + '.loss'])) <= 1e-5 * abs(float(g[tag + '.loss']))
    for k, v in c.items():
        ref = g[f'{tag}.{k}']
        if v.grad is None:
            assert np.abs(ref).max() == 0, k
            continue
        assert v.grad.shape == v.shape
        grad_close(v.grad, ref, f'{stage}/{tag} d/d {k}')
    for name, p in dec.named_parameters():
        ref = g[f'{tag}.sd.{name}']
        if p.grad is None:
            assert np.abs(ref).max() == 0, name
            continue
        grad_close(p.grad, ref, f'{stage}/{tag} d/d {name}', mode=mode)


@pytest.mark.parametrize('mode', MODES)
def test_gradients_second_seed_64_samples_vs_oracle(mini, mode, monkeypatch):
    """S = 64 (benchmark sampling), other weights, more rays, against the oracle's autograd."""
    monkeypatch.setenv('ADFP_MATH', mode)
    sd = O.random_state_dict(seed=17)
    sc = synthetic.mini_scene()
    rays = synthetic.make_ray_batch(sc, 300, seed=8, poses=3)
    loss, c, dec = run(mini, 'color', True, sd=sd, n_samples=48, n_surface=16, rays=rays)
    c_or = {k: v.clone().requires_grad_(True) for k, v in mini.c.items()}
    sd_or = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    d2, u2, col2, w2 = O.render_batch_ray(sd_or, c_or, rays[1], rays[0], mini.tsdf_volume, mini.tsdf_bnds, mini.bound,
                                          'color', rays[2], 48, 16)
    loss2 = O.mapper_loss(d2, col2, w2, rays[2], rays[3], 'color', True)
    loss2.backward()
    assert abs(loss.item() - loss2.item()) <= 1e-5 * abs(loss2.item())
    for k in c:
        grad_close(c[k].grad, c_or[k].grad, k)
    for name, p in dec.named_parameters():
        ref = sd_or[name].grad if sd_or[name].grad is not None else torch.zeros_like(sd_or[name])
        grad_close(p.grad if p.grad is not None else torch.zeros_like(p), ref, name, mode=mode)


def test_uncertainty_cotangent_vs_oracle(mini):
    """The Tracker's loss divides by sqrt(uncertainty) (src/Tracker.py:116-121); the Mapper never
    differentiates it, so it is pinned here against the oracle."""
    dec = A.DF()
    dec.load_state_dict(mini.sd)
    dec.bound = mini.bound
    dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(mini.n_samples, mini.n_surface), None, mini)
    c = {k: v.to(DEV).clone().requires_grad_(True) for k, v in mini.c.items()}
    d, u, col, w = rend.render_batch_ray(c, dec, mini.rays_d.to(DEV), mini.rays_o.to(DEV), DEV,
                                         mini.tsdf_volume.to(DEV), mini.tsdf_bnds.to(DEV), 'high', gt_depth=mini.gt_depth.to(DEV))
    (u.sum() * 3.0 + d.sum()).backward()
    c_or = {k: v.clone().requires_grad_(True) for k, v in mini.c.items()}
    d2, u2, col2, w2 = O.render_batch_ray(mini.sd, c_or, mini.rays_d, mini.rays_o, mini.tsdf_volume, mini.tsdf_bnds,
                                          mini.bound, 'high', mini.gt_depth, mini.n_samples, mini.n_surface)
    (u2.sum() * 3.0 + d2.sum()).backward()
    for k in ('grid_low', 'grid_high'):
        grad_close(c[k].grad, c_or[k].grad, k)


def test_frozen_decoders_and_index_put_grids(mini):
    """Mapper configuration: low decoder never optimised, high frozen (fix_high), grids are autograd
    non-leafs made by index_put (src/Mapper.py:364-388); gradients must arrive at the masked leaf."""
    dec = A.DF()
    dec.load_state_dict(mini.sd)
    dec.bound = mini.bound
    dec = dec.to(DEV)
    for p in list(dec.low_decoder.parameters()) + list(dec.high_decoder.parameters()):
        p.requires_grad_(False)
    rend = A.Renderer(make_cfg(mini.n_samples, mini.n_surface), None, mini)
    c = {}
    leaves = {}
    for k, v in mini.c.items():
        val = v.to(DEV).clone()
        mask = torch.rand(val.shape, device=DEV) < 0.6
        leaf = val[mask].clone().requires_grad_(True)
        val[mask] = leaf
        c[k], leaves[k] = val, (leaf, mask)
    gd, gc = mini.gt_depth.to(DEV), mini.gt_color.to(DEV)
    d, u, col, w = rend.render_batch_ray(c, dec, mini.rays_d.to(DEV), mini.rays_o.to(DEV), DEV,
                                         mini.tsdf_volume.to(DEV), mini.tsdf_bnds.to(DEV), 'color', gt_depth=gd)
    mapper_loss(d, col, w, gd, gc, 'color', False).backward()
    g = mini.golden('color')
    for k, (leaf, mask) in leaves.items():
        ref = torch.from_numpy(g[f'g.{k}'])[mask.cpu()]
        grad_close(leaf.grad, ref, f'masked leaf of {k}')
    assert all(p.grad is None for p in dec.low_decoder.parameters())
    assert all(p.grad is None for p in dec.high_decoder.parameters())
    assert all(p.grad is not None for p in dec.color_decoder.parameters())
    grad_close(dec.mlp.pts_linears[2].weight.grad, g['g.sd.mlp.pts_linears.2.weight'], 'mlp W2')


def test_adam_steps_reduce_loss(mini):
    """A few Mapper-style iterations (fresh Adam, stage color) run and reduce the loss."""
    dec = A.DF()
    dec.load_state_dict(mini.sd)
    dec.bound = mini.bound
    dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(mini.n_samples, mini.n_surface), None, mini)
    c = {k: v.to(DEV).clone().requires_grad_(True) for k, v in mini.c.items()}
    opt = torch.optim.Adam([{'params': list(dec.color_decoder.parameters()) + list(dec.mlp.parameters()), 'lr': 0.005},
                            {'params': list(c.values()), 'lr': 0.005}])
    gd, gc = mini.gt_depth.to(DEV), mini.gt_color.to(DEV)
    tsdf, bnds = mini.tsdf_volume.to(DEV), mini.tsdf_bnds.to(DEV)
    losses = []
    for it in range(12):
        opt.zero_grad()
        d, u, col, w = rend.render_batch_ray(c, dec, mini.rays_d.to(DEV), mini.rays_o.to(DEV), DEV, tsdf, bnds, 'color',
                                             gt_depth=gd)
        loss = mapper_loss(d, col, w, gd, gc, 'color', False)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < 0.8 * losses[0], losses


def tracker_loss(depth, unc, color, gt_depth, gt_color):
    unc = unc.detach()
    tmp = torch.abs(gt_depth - depth) / torch.sqrt(unc + 1e-10)
    mask = (tmp < 10 * tmp.median()) & (gt_depth > 0)
    return (torch.abs(gt_depth - depth) / torch.sqrt(unc + 1e-10))[mask].sum() + 0.5 * torch.abs(gt_color - color)[mask].sum()


def test_tracker_ray_gradients_vs_reference_golden(mini):
    """Camera tracking (src/Tracker.py:112-133): gradients w.r.t. rays_o / rays_d through the trilinear
    coordinates of the three feature grids and of the TSDF and through sin(p @ B), against the reference's."""
    g = mini.golden('tracker')
    dec = A.DF()
    dec.load_state_dict(mini.sd)
    dec.bound = mini.bound
    dec = dec.to(DEV)
    for p in dec.parameters():
        p.requires_grad_(False)
    rend = A.Renderer(make_cfg(mini.n_samples, mini.n_surface), None, mini)
    c = to_dev(mini.c, DEV)
    ro = mini.rays_o.to(DEV).clone().requires_grad_(True)
    rd = mini.rays_d.to(DEV).clone().requires_grad_(True)
    gd, gc = mini.gt_depth.to(DEV), mini.gt_color.to(DEV)
    d, u, col, w = rend.render_batch_ray(c, dec, rd, ro, DEV, mini.tsdf_volume.to(DEV), mini.tsdf_bnds.to(DEV), 'color',
                                         gt_depth=gd)
    loss = tracker_loss(d, u, col, gd, gc)
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= 2e-5 * abs(float(g['loss']))
    grad_close(ro.grad, g['g_rays_o'], 'd/d rays_o', tol=5e-4)
    grad_close(rd.grad, g['g_rays_d'], 'd/d rays_d', tol=5e-4)


@pytest.mark.parametrize('mode', MODES)
def test_tracker_ray_gradients_along_the_kernels_relu_decisions(mini, mode, monkeypatch):
    """The same ray gradients (d/d position through the trilinear coordinates of the grids and of the TSDF and through
    sin(p @ B), reduced per ray) against the oracle's autograd along the kernels' ReLU decisions: every element within
    conftest.TIGHT_GRAD_TOL x scale, on the f16-split PGRAD kernels and on the exact ones."""
    monkeypatch.setenv('ADFP_MATH', mode)
    dec = A.DF()
    dec.load_state_dict(mini.sd)
    dec.bound = mini.bound
    dec = dec.to(DEV)
    for p in dec.parameters():
        p.requires_grad_(False)
    rend = A.Renderer(make_cfg(mini.n_samples, mini.n_surface), None, mini)
    cap = ReluCapture(rend)
    c = to_dev(mini.c, DEV)
    ro = mini.rays_o.to(DEV).clone().requires_grad_(True)
    rd = mini.rays_d.to(DEV).clone().requires_grad_(True)
    gd, gc = mini.gt_depth.to(DEV), mini.gt_color.to(DEV)
    d, u, col, w = rend.render_batch_ray(c, dec, rd, ro, DEV, mini.tsdf_volume.to(DEV), mini.tsdf_bnds.to(DEV), 'color', gt_depth=gd)
    loss = tracker_loss(d, u, col, gd, gc)
    loss.backward()
    ro_o, rd_o = mini.rays_o.clone().requires_grad_(True), mini.rays_d.clone().requires_grad_(True)
    O.reset_relu_flips()
    d2, u2, col2, w2 = O.render_batch_ray(mini.sd, mini.c, rd_o, ro_o, mini.tsdf_volume, mini.tsdf_bnds, mini.bound, 'color',
                                          mini.gt_depth, mini.n_samples, mini.n_surface, relu_masks=cap.masks('color'))
    assert_forced_decisions_are_boundary_units(dict(O.RELU_FLIPS))
    loss2 = O.tracker_loss(d2, u2, col2, mini.gt_depth, mini.gt_color)
    loss2.backward()
    assert abs(loss.item() - loss2.item()) <= 2e-5 * abs(loss2.item())
    assert_grad_tight(ro.grad, ro_o.grad, 'd/d rays_o', mode)
    assert_grad_tight(rd.grad, rd_o.grad, 'd/d rays_d', mode)


def test_pose_gradient_through_get_rays_from_uv(mini):
    """End to end like Tracker.optimize_cam_in_batch: a c2w matrix that requires grad -> rays -> render -> loss;
    the pose gradient equals the oracle's."""
    from attentive_dfprior_amd import common
    dec = A.DF()
    dec.load_state_dict(mini.sd)
    dec.bound = mini.bound
    dec = dec.to(DEV)
    for p in dec.parameters():
        p.requires_grad_(False)
    rend = A.Renderer(make_cfg(mini.n_samples, mini.n_surface), None, mini)
    c = to_dev(mini.c, DEV)
    gen = torch.Generator().manual_seed(3)
    i = torch.randint(4, mini.W - 4, (120,), generator=gen).float()
    j = torch.randint(4, mini.H - 4, (120,), generator=gen).float()
    gd = mini.depth_img[j.long(), i.long()]
    gcol = torch.rand(120, 3, generator=gen)

    c2w = mini.c2w.to(DEV).clone().requires_grad_(True)
    ro, rd = common.get_rays_from_uv(i.to(DEV), j.to(DEV), c2w, mini.H, mini.W, mini.fx, mini.fy, mini.cx, mini.cy, DEV)
    d, u, col, w = rend.render_batch_ray(c, dec, rd, ro, DEV, mini.tsdf_volume.to(DEV), mini.tsdf_bnds.to(DEV), 'color',
                                         gt_depth=gd.to(DEV))
    tracker_loss(d, u, col, gd.to(DEV), gcol.to(DEV)).backward()

    c2w_o = mini.c2w.clone().requires_grad_(True)
    ro2, rd2 = O.get_rays_from_uv(i, j, c2w_o, mini.fx, mini.fy, mini.cx, mini.cy)
    d2, u2, col2, w2 = O.render_batch_ray(mini.sd, mini.c, rd2, ro2, mini.tsdf_volume, mini.tsdf_bnds, mini.bound, 'color',
                                          gd, mini.n_samples, mini.n_surface)
    O.tracker_loss(d2, u2, col2, gd, gcol).backward()
    grad_close(c2w.grad[:3], c2w_o.grad[:3], 'd/d c2w', tol=5e-4)


# --------------------------------------------------------------------------- autograd through eval_points / DF.forward
@pytest.mark.parametrize('stage', O.STAGES)
@pytest.mark.parametrize('via', ['eval_points', 'df_forward'])
def test_eval_points_is_autograd_transparent(mini, stage, via):
    """The reference's Renderer.eval_points / DF.forward are plain torch ops (src/utils/Renderer.py:27-71,
    src/conv_onet/models/decoder.py:307-353): gradients reach the query points, the grids and the decoder parameters.  Here they
    run through adfp_eval_points_train / adfp_eval_points_backward and must match the oracle's autograd, including the
    bound rule (occupancy 100 outside the bound: no gradient) that eval_points applies and DF.forward does not."""
    g = torch.Generator().manual_seed(12)
    lo, hi = mini.bound[:, 0], mini.bound[:, 1]
    pts = lo + (hi - lo) * (torch.rand(600, 3, generator=g, dtype=torch.float64) * 1.1 - 0.05)     # some outside the bound
    wr, ww = torch.randn(600, 4, generator=g), torch.randn(600, generator=g)
    # oracle
    c_o = {k: v.clone().requires_grad_(True) for k, v in mini.c.items()}
    sd_o = {k: v.clone().requires_grad_(True) for k, v in mini.sd.items()}
    p_o = pts.clone().requires_grad_(True)
    if via == 'eval_points':
        raw_o, w_o = O.eval_points(sd_o, p_o, c_o, mini.tsdf_volume, mini.tsdf_bnds, mini.bound, stage)
    else:
        raw_o, w_o = O.df_forward(sd_o, p_o, c_o, mini.tsdf_volume, mini.tsdf_bnds, mini.bound, stage)
    ((raw_o * wr).sum() + (w_o * ww).sum()).backward()
    # product
    dec = A.DF()
    dec.load_state_dict(mini.sd)
    dec.bound = mini.bound
    dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(mini.n_samples, mini.n_surface), None, mini)
    c = {k: v.to(DEV).clone().requires_grad_(True) for k, v in mini.c.items()}
    p = pts.to(DEV).requires_grad_(True)
    tsdf, tb = mini.tsdf_volume.to(DEV), mini.tsdf_bnds.to(DEV)
    if via == 'eval_points':
        raw, w = rend.eval_points(p, dec, tsdf, tb, c, stage, DEV)
    else:
        raw, w = dec(p.unsqueeze(0), c_grid=c, tsdf_volume=tsdf, tsdf_bnds=tb, stage=stage)
    assert raw.requires_grad
    grad_close(raw, raw_o.detach(), 'raw', 1e-4)
    ((raw * wr.to(DEV)).sum() + (w * ww.to(DEV)).sum()).backward()
    assert p.grad.dtype == torch.float64
    grad_close(p.grad, p_o.grad, 'd/d points', 5e-4)
    used = {'low': ('grid_low',), 'high': ('grid_low', 'grid_high'), 'color': ('grid_low', 'grid_high', 'grid_color')}[stage]
    for k in c:
        if k in used:
            grad_close(c[k].grad, c_o[k].grad, k)
        else:
            assert c[k].grad is None or float(c[k].grad.abs().max()) == 0.0
    for name, prm in dec.named_parameters():
        ref = sd_o[name].grad
        if ref is None or float(ref.abs().max()) == 0.0:
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, name
        else:
            grad_close(prm.grad, ref, name)


def _grads_of_case(mini, monkeypatch, mode, no_masks=False):
    """Gradients of the second-seed case (300 rays x 64 samples, all networks trainable) under ADFP_MATH=mode; no_masks: the
    training forward leaves no ReLU masks, so the f16-split forward is followed by the EXACT backward."""
    from attentive_dfprior_amd import engine
    monkeypatch.setenv('ADFP_MATH', mode)
    if no_masks:
        orig = engine.Engine.train_state

        def without_masks(P, stage, dev, decoders, need_flat=None, **kw):
            monkeypatch.setenv('ADFP_MATH', 'f32')
            try:
                return orig(P, stage, dev, decoders, need_flat, **kw)
            finally:
                monkeypatch.setenv('ADFP_MATH', mode)
        monkeypatch.setattr(engine.Engine, 'train_state', staticmethod(without_masks))
    sd = O.random_state_dict(seed=17)
    rays = synthetic.make_ray_batch(synthetic.mini_scene(), 300, seed=8, poses=3)
    loss, c, dec = run(mini, 'color', True, sd=sd, n_samples=48, n_surface=16, rays=rays)
    out = {k: v.grad.detach().clone() for k, v in c.items()}
    out.update({n: p.grad.detach().clone() for n, p in dec.named_parameters() if p.grad is not None})
    return out


def test_f16_split_backward_against_the_exact_backward(mini, monkeypatch):
    """Same f16-split forward, then the two backwards: k_decode_bwd_h / k_outer_h / k_scatter_sorted (masks and layer inputs from
    the forward, f16 MFMA with operand split, sorted scatter) against k_decode_bwd / k_outer_lds (recompute, f32 MFMA, in-kernel
    scatter).  Measured: every element agrees to ~3e-7 of its tensor's scale except the row of a unit whose ReLU the two
    forwards decide differently (tools/diag_bwd.py)."""
    exact = _grads_of_case(mini, monkeypatch, 'f16x3', no_masks=True)
    monkeypatch.undo()
    split = _grads_of_case(mini, monkeypatch, 'f16x3')
    assert set(exact) == set(split)
    tight = 0
    for k in exact:
        a, b = split[k].double().cpu(), exact[k].double().cpu()
        scale = b.abs().max().clamp_min(1e-30)
        err = (a - b).abs() / scale
        tight += int((err <= 5e-6).all())
        grad_close(split[k], exact[k], f'f16-split vs exact backward: {k}', tol=1e-4)
    # How many tensors a boundary sample touches depends on WHERE it sits: a flipped unit of the attention MLP changes d/d(high +
    # low) of its sample and with it every tensor of the low and high decoders (two thirds of all tensors, profiles/
    # r03_diag_backward_modes.txt); a flip in a decoder's first layer touches one row.  So every tensor is held to the cap /
    # Frobenius limits above, and the tensors no flip reaches -- at least the colour decoder's, a fifth of all -- must agree to
    # fp32 rounding, which is what shows that the f16-split backward's ARITHMETIC is fp32-grade.
    assert tight >= 0.2 * len(exact), f'only {tight} of {len(exact)} tensors agree to 5e-6'


def test_sorted_scatter_equals_cached_scatter(mini):
    """Grid gradients through k_scatter_sorted (radix sort by cell, run-length sums) and through the in-kernel write-combining
    scatter (adfp_backward_args.options = ADFP_BWD_SCATTER_IN_KERNEL, a field of the descriptor the host passes)."""
    from attentive_dfprior_amd import _lib
    rays = synthetic.make_ray_batch(synthetic.mini_scene(), 300, seed=8, poses=3)
    got = {}
    for name, opt in (('sorted', 0), ('cache', _lib.BWD_SCATTER_IN_KERNEL)):
        loss, c, dec = run(mini, 'color', True, sd=O.random_state_dict(seed=17), n_samples=48, n_surface=16, rays=rays, bwd_options=opt)
        got[name] = {k: v.grad.cpu() for k, v in c.items()}
    for k in got['sorted']:
        assert_close_scale(got['sorted'][k], got['cache'][k], 2e-6, f'{k}: sorted vs cached scatter')


@pytest.mark.parametrize('n_rays', [3, 700])
def test_side_lane_changes_nothing_but_the_schedule(mini, n_rays):
    """adfp_backward_args.side_stream (round 6): the spatial sort on a second stream beside the backward kernels, the call's
    persistent kernels on ADFP_SIDE_CU_RESERVE fewer compute units.  Against the one-stream backward: the same gradients up to the
    summation order (fewer workgroups = fewer partial sums; float atomics in the scatter), every stage, and the calls are
    ordered -- a gradient read right after backward() on the caller's stream is complete."""
    rays = synthetic.make_ray_batch(synthetic.mini_scene(), n_rays, seed=8, poses=3)
    for stage in ('low', 'high', 'color'):
        got = {}
        for name, lane in (('side', True), ('one_stream', False)):
            loss, c, dec = run(mini, stage, stage != 'color', sd=O.random_state_dict(seed=17), n_samples=48, n_surface=16, rays=rays, side_lane=lane)
            got[name] = {k: v.grad.clone() for k, v in c.items() if v.grad is not None}              # read on the caller's stream, no synchronize
            got[name].update({n: p.grad.detach().clone() for n, p in dec.named_parameters() if p.grad is not None})
        assert set(got['side']) == set(got['one_stream']) and len(got['side']) >= 1
        for k in got['side']:
            assert torch.isfinite(got['side'][k]).all(), k
            assert_close_scale(got['side'][k], got['one_stream'][k], 3e-6, f'{stage}, {n_rays} rays: {k}: side lane vs one stream')


@pytest.mark.parametrize('stage', ['low', 'color'])
def test_fused_weight_gradients_equal_the_staged_path(mini, stage):
    """The 32-channel decoders' weight gradients from inside the chain kernel -- the role-split kernel at two waves per SIMD
    (k_decode_bwd_roles, the default: three kinds of workgroup, each with a third of the gradient blocks) and round 3-4's one-wave
    kernel (k_decode_bwd_fused, ADFP_BWD_FUSED_ONE_WAVE) -- against the staged two-kernel path (ADFP_BWD_STAGED_WGRAD: cotangent
    blocks written per point, k_outer_h): the same forward state, the same 3-product split -- only the summation order over the
    points differs."""
    from attentive_dfprior_amd import _lib
    rays = synthetic.make_ray_batch(synthetic.mini_scene(), 700, seed=5, poses=3)
    got = {}
    for name, opt in (('fused', 0), ('one_wave', _lib.BWD_FUSED_ONE_WAVE), ('staged', _lib.BWD_STAGED_WGRAD)):
        loss, c, dec = run(mini, stage, stage != 'color', sd=O.random_state_dict(seed=23), n_samples=48, n_surface=16, rays=rays, bwd_options=opt)
        got[name] = {n: p.grad.detach().cpu().clone() for n, p in dec.named_parameters() if p.grad is not None}
        got[name].update({k: v.grad.cpu() for k, v in c.items() if v.grad is not None})
    assert set(got['fused']) == set(got['staged']) and any(k.startswith('low_decoder') for k in got['fused'])
    for k in got['fused']:
        assert_close_scale(got['fused'][k], got['staged'][k], 3e-6, f'{k}: role-split kernel vs staged weight gradients')
        assert_close_scale(got['one_wave'][k], got['staged'][k], 3e-6, f'{k}: one-wave kernel vs staged weight gradients')
        if stage == 'color' and k.startswith('color_decoder'):
            assert got['fused'][k].abs().max() > 0, k


@pytest.mark.parametrize('n_rays', [1, 5, 67, 700])
def test_fused_weight_gradients_on_ragged_point_counts(mini, n_rays):
    """Point counts that do not fill a 32-point tile, a wave or a workgroup (1 x 40 = 40 points: one partial second tile; 5 x 40 =
    200; 67 x 40 = 2 680 = 83.75 tiles): the fused kernel's idle waves, clamped DMA rows and zeroed tail lanes."""
    from attentive_dfprior_amd import _lib
    rays = synthetic.make_ray_batch(synthetic.mini_scene(), n_rays, seed=9, poses=1, zero_frac=0.0)
    got = {}
    for name, opt in (('fused', 0), ('one_wave', _lib.BWD_FUSED_ONE_WAVE), ('staged', _lib.BWD_STAGED_WGRAD)):
        loss, c, dec = run(mini, 'color', False, sd=O.random_state_dict(seed=29), n_samples=24, n_surface=16, rays=rays, bwd_options=opt)
        got[name] = {n: p.grad.detach().cpu().clone() for n, p in dec.named_parameters() if p.grad is not None}
        got[name].update({k: v.grad.cpu() for k, v in c.items() if v.grad is not None})
    for k in got['fused']:
        assert torch.isfinite(got['fused'][k]).all(), k
        assert_close_scale(got['fused'][k], got['staged'][k], 3e-6, f'{k}: role-split kernel vs staged, {n_rays} rays')
        assert_close_scale(got['one_wave'][k], got['staged'][k], 3e-6, f'{k}: one-wave kernel vs staged, {n_rays} rays')


def test_a_swapped_pair_of_weight_rows_fails_the_gradient_comparison(mini, monkeypatch):
    """Negative control of the comparison itself: the backward's flat colour-decoder gradient comes back with two rows of
    pts_linears.1.weight exchanged (what an indexing bug in k_outer_h's write-out would produce) -- the golden comparison
    must fail, in both math modes."""
    from attentive_dfprior_amd import engine
    L = engine.lib()
    orig = engine.Engine.render_backward

    def swapped(self, *a, **k):
        grids, flats, rays = orig(self, *a, **k)
        if 'color' in flats:
            names = [n for n, _ in A.DF().color_decoder.named_parameters()]
            shapes = [tuple(p.shape) for _, p in A.DF().color_decoder.named_parameters()]
            off = 0
            for n, shp in zip(names, shapes):
                cnt = int(np.prod(shp))
                if n == 'pts_linears.1.weight':
                    w = flats['color'][off:off + cnt].view(shp)
                    r3, r5 = w[3].clone(), w[5].clone()
                    w[3], w[5] = r5, r3
                off += cnt
            assert off == L.adfp_decoder_flat_floats(2)
        return grids, flats, rays
    monkeypatch.setattr(engine.Engine, 'render_backward', swapped)
    for mode in MODES:
        monkeypatch.setenv('ADFP_MATH', mode)
        g = mini.golden('color')
        loss, c, dec = run(mini, 'color', False)
        p = dict(dec.named_parameters())['color_decoder.pts_linears.1.weight']
        with pytest.raises(AssertionError):
            grad_close(p.grad, g['g.sd.color_decoder.pts_linears.1.weight'], 'swapped color_decoder.pts_linears.1.weight', mode=mode)
        q = dict(dec.named_parameters())['color_decoder.pts_linears.2.weight']           # an untouched tensor still passes
        grad_close(q.grad, g['g.sd.color_decoder.pts_linears.2.weight'], 'color_decoder.pts_linears.2.weight', mode=mode)
