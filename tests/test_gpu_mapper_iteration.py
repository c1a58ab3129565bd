"""GPU: mapping.MapperIteration -- one Mapper iteration (pre-filter, render, loss, backward, Adam) as a fixed kernel
sequence replayed from a HIP graph -- against the reference-shaped path it replaces: common.filter_rays_in_bound
(boolean compaction), Renderer.render_batch_ray under autograd, the loss written with torch ops (src/Mapper.py:457-469),
loss.backward(), torch.optim.Adam on the decoder parameters and MaskedGridAdam on the grids.  Same rays, same start,
several iterations through the low -> high -> color stages with the warm-up term: the parameters must agree."""
import copy

import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import common, mapping, synthetic
from oracle import adfp_oracle as O
from conftest import make_cfg, to_dev, assert_close, assert_adam_trajectory

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
STAGE_LR = {'low': dict(low=0.1, high=0.0, color=0.0, decoders=0.0, mlp=0.0),
            'high': dict(low=0.005, high=0.005, color=0.0, decoders=0.0, mlp=0.005),
            'color': dict(low=0.005, high=0.005, color=0.005, decoders=0.005, mlp=0.005)}
SCHEDULE = [('low', False), ('low', False), ('high', True), ('high', False), ('color', False), ('color', False)]


def setup():
    sc = synthetic.mini_scene()
    sd = O.random_state_dict(seed=3)
    dec = A.DF()
    dec.load_state_dict(sd)
    dec.bound = sc.bound
    dec = dec.to(DEV)
    for p in list(dec.low_decoder.parameters()) + list(dec.high_decoder.parameters()):
        p.requires_grad_(False)
    rend = A.Renderer(make_cfg(32, 16), None, sc)
    ro, rd, gd, gc = synthetic.make_ray_batch(sc, 700, seed=9, zero_frac=0.1)
    gd = gd * (0.6 + 1.2 * torch.rand(gd.shape, generator=torch.Generator().manual_seed(1)))     # some beyond the box: dropped
    rd[5, 0] = 0.0
    c2w = sc.default_c2w(yaw=0.7, pitch=0.1)
    masks = {k: mapping.frustum_mask(c2w, tuple(v.shape[2:]), sc.depth_image(c2w).to(DEV), sc.bound, sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy)
             for k, v in sc.c.items()}
    return sc, dec, rend, [t.to(DEV) for t in (ro, rd, gd, gc)], masks


def reference_path(sc, dec, rend, rays, masks):
    ro, rd, gd, gc = rays
    grids = {k: v.clone().to(DEV).requires_grad_(True) for k, v in sc.c.items()}
    opt_g = mapping.MaskedGridAdam(grids, masks)
    opt = torch.optim.Adam([{'params': list(dec.color_decoder.parameters()), 'lr': 0}, {'params': list(dec.mlp.parameters()), 'lr': 0}])
    tsdf, tb, bound = sc.tsdf_volume.to(DEV), sc.tsdf_bnds.to(DEV), sc.bound.to(DEV)
    losses = []
    for stage, warm in SCHEDULE:
        lr = STAGE_LR[stage]
        opt.param_groups[0]['lr'], opt.param_groups[1]['lr'] = lr['decoders'], lr['mlp']
        opt.zero_grad(); opt_g.zero_grad()
        o, d, z, c = common.filter_rays_in_bound(ro, rd, gd, gc, bound)
        depth, unc, col, w = rend.render_batch_ray(grids, dec, d, o, DEV, tsdf, tb, stage, z)
        m = z > 0
        loss = torch.abs(z[m] - depth[m]).sum()
        if warm:
            loss = loss + torch.abs(w - 1.0).sum()
        if stage == 'color':
            loss = loss + 0.2 * torch.abs(c - col).sum()
        loss.backward()
        opt.step()
        opt_g.step({'grid_low': lr['low'], 'grid_high': lr['high'], 'grid_color': lr['color']})
        losses.append(float(loss))
    return {k: v.detach().clone() for k, v in grids.items()}, {n: p.detach().clone() for n, p in dec.named_parameters()}, losses


@pytest.mark.parametrize('use_graph', [False, True])
def test_fused_iteration_follows_the_reference_shaped_path(use_graph):
    sc, dec, rend, rays, masks = setup()
    dec2 = copy.deepcopy(dec)
    g_ref, p_ref, l_ref = reference_path(sc, dec, rend, rays, masks)

    grids = {k: v.clone().to(DEV) for k, v in sc.c.items()}
    rend2 = A.Renderer(make_cfg(32, 16), None, sc)
    it = mapping.MapperIteration(rend2, dec2, grids, masks, sc.tsdf_volume.to(DEV), sc.tsdf_bnds.to(DEV), STAGE_LR, use_graph=use_graph)
    sd_keys = list(dec2.state_dict().keys())
    losses = [float(it.step(*rays, stage, warm)) for stage, warm in SCHEDULE]     # graph mode: a stage's second visit replays
    assert list(dec2.state_dict().keys()) == sd_keys          # flattening the parameters left the module's interface alone
    for a, b in zip(losses, l_ref):
        assert abs(a - b) <= 1e-5 * abs(b), (losses, l_ref)
    for k in g_ref:
        assert_adam_trajectory(grids[k], g_ref[k], 0.1 if k == 'grid_low' else 0.005, len(SCHEDULE), f'{k} after {len(SCHEDULE)} iterations')
        outside = ~masks[k].to(DEV)
        assert torch.equal(grids[k][0, :, outside], sc.c[k].to(DEV)[0, :, outside])     # untouched outside the frustum mask
    for n, p in dec2.named_parameters():
        assert_adam_trajectory(p, p_ref[n], 0.005, len(SCHEDULE), n)
    # the drop-in objects keep working after the fused iterations (version bumps invalidate the layout caches)
    with torch.no_grad():
        d1 = rend2.render_batch_ray(grids, dec2, rays[1], rays[0], DEV, sc.tsdf_volume.to(DEV), sc.tsdf_bnds.to(DEV), 'color', rays[2])[0]
        d2 = rend.render_batch_ray({k: v for k, v in g_ref.items()}, dec, rays[1], rays[0], DEV, sc.tsdf_volume.to(DEV), sc.tsdf_bnds.to(DEV),
                                   'color', rays[2])[0]
    ok = torch.isfinite(d2)
    assert_close(d1[ok], d2[ok], 1e-3, 'render after training, fused vs reference-shaped path')


def test_graph_replays_with_new_rays():
    """The captured sequence reads its rays from static buffers: a replay with different rays must equal the unfused
    iteration on those rays."""
    sc, dec, rend, rays, masks = setup()
    dec_b = copy.deepcopy(dec)
    tsdf, tb = sc.tsdf_volume.to(DEV), sc.tsdf_bnds.to(DEV)
    ga = {k: v.clone().to(DEV) for k, v in sc.c.items()}
    gb = {k: v.clone().to(DEV) for k, v in sc.c.items()}
    it_a = mapping.MapperIteration(A.Renderer(make_cfg(32, 16), None, sc), dec, ga, masks, tsdf, tb, STAGE_LR, use_graph=True)
    it_b = mapping.MapperIteration(A.Renderer(make_cfg(32, 16), None, sc), dec_b, gb, masks, tsdf, tb, STAGE_LR, use_graph=False)
    for seed in (1, 2, 3):
        ro, rd, gd, gc = [t.to(DEV) for t in synthetic.make_ray_batch(sc, 700, seed=seed)]
        la = float(it_a.step(ro, rd, gd, gc, 'color'))
        lb = float(it_b.step(ro, rd, gd, gc, 'color'))
        assert abs(la - lb) <= 1e-6 * abs(lb)
    for k in ga:
        assert_adam_trajectory(ga[k], gb[k], 0.005, 3, k)
    for (n, p), (_, q) in zip(dec.named_parameters(), dec_b.named_parameters()):
        assert_adam_trajectory(p, q, 0.005, 3, n)


def test_iteration_reads_the_corner_block_tsdf_and_follows_a_rewritten_volume():
    """The iteration reads the corner-block copy of the TSDF volume (Engine.tsdf_blocks) -- the same values as the volume itself, so
    the losses are the plain-volume iteration's bit for bit -- and when somebody writes the volume in place (TSDF fusion between
    keyframes) the copy is re-laid at its address, which the captured graph keeps using."""
    sc, dec, rend, rays, masks = setup()
    dec_b, dec_c = copy.deepcopy(dec), copy.deepcopy(dec)
    tsdf, tb = sc.tsdf_volume.to(DEV).clone(), sc.tsdf_bnds.to(DEV)
    tsdf_b = tsdf.clone()
    ga, gb, gc_ = ({k: v.clone().to(DEV) for k, v in sc.c.items()} for _ in range(3))
    rend_plain = A.Renderer(make_cfg(32, 16), None, sc)
    rend_plain.tsdf_blocks = False
    it_a = mapping.MapperIteration(A.Renderer(make_cfg(32, 16), None, sc), dec, ga, masks, tsdf, tb, STAGE_LR, use_graph=True)
    it_b = mapping.MapperIteration(rend_plain, dec_b, gb, masks, tsdf_b, tb, STAGE_LR, use_graph=False)
    assert it_a._cb is not None and it_b._cb is None
    address = it_a._cb.data_ptr()
    for k in range(2):
        la, lb = float(it_a.step(*rays, 'color')), float(it_b.step(*rays, 'color'))
        assert la == lb if k == 0 else abs(la - lb) <= 1e-6 * abs(lb), (k, la, lb)      # (from the second step on the atomics' order tells)
    # the volume changes under both (in place): every value moved, the band with it
    for t in (tsdf, tsdf_b):
        t.mul_(0.5).add_(0.01)
    for k in range(2):
        la, lb = float(it_a.step(*rays, 'color')), float(it_b.step(*rays, 'color'))
        assert abs(la - lb) <= 1e-6 * abs(lb), ('after the rewrite', k, la, lb)
    assert it_a._cb.data_ptr() == address
    # and differs from an iteration that still saw the old volume
    it_c = mapping.MapperIteration(rend_plain, dec_c, gc_, masks, sc.tsdf_volume.to(DEV), tb, STAGE_LR, use_graph=False)
    lc = [float(it_c.step(*rays, 'color')) for _ in range(4)]
    assert abs(lc[3] - la) > 1e-3 * abs(la)


def test_eager_calls_between_graph_replays_see_valid_caches():
    """Three stages captured into one shared graph pool, replayed out of capture order, and an eager render with the SAME
    decoders / grids in between (what Visualizer.vis, the Tracker and the Mesher do while the Mapper iterates): the eager
    call's cached weight images and channels-last grids must not be blocks that another graph's replay writes.  Compared
    against a fresh DF / Renderer holding copies of the current parameters."""
    sc, dec, rend, rays, masks = setup()
    tsdf, tb = sc.tsdf_volume.to(DEV), sc.tsdf_bnds.to(DEV)
    grids = {k: v.clone().to(DEV) for k, v in sc.c.items()}
    it = mapping.MapperIteration(rend, dec, grids, masks, tsdf, tb, STAGE_LR, use_graph=True)
    for stage in ('low', 'high', 'color'):
        it.step(*rays, stage)                                   # captures (and replays once) each stage's graph
    ws_before = rend._engine._ws
    for stage in ('low', 'color', 'high', 'low', 'color'):      # any order
        it.step(*rays, stage)
        with torch.no_grad():
            got = rend.render_batch_ray(grids, dec, rays[1], rays[0], DEV, tsdf, tb, 'color', rays[2])
            fresh_dec = A.DF()
            fresh_dec.load_state_dict({k: v.detach().clone() for k, v in dec.state_dict().items()})
            fresh_dec.bound = sc.bound
            fresh_dec = fresh_dec.to(DEV)
            want = A.Renderer(make_cfg(32, 16), None, sc).render_batch_ray({k: v.clone() for k, v in grids.items()}, fresh_dec, rays[1], rays[0],
                                                                         DEV, tsdf, tb, 'color', rays[2])
        for a, b in zip(got, want):
            ok = torch.isfinite(b)
            assert torch.equal(a[ok], b[ok]), f'eager render after replaying stage {stage} differs from a fresh engine'
    # no cache entry of the shared objects may point into the graphs' pool: the trained nets' images and the grids' copies are
    # rebuilt eagerly after every replay (their versions were bumped), the frozen nets' images were packed before the captures
    assert rend._engine._ws is ws_before or rend._engine._ws is not None
    # a big eager render grows (reallocates) the engine's workspace; the graphs own theirs and must keep replaying correctly
    big = [t.to(DEV) for t in synthetic.make_ray_batch(sc, 20000, seed=4)]
    with torch.no_grad():
        rend.render_batch_ray(grids, dec, big[1], big[0], DEV, tsdf, tb, 'color', big[2])
    junk = [torch.full((1 << 20,), float('nan'), device=DEV) for _ in range(8)]          # whatever the old workspace block becomes
    dec_b = copy.deepcopy(dec)
    gb = {k: v.clone() for k, v in grids.items()}
    it_b = mapping.MapperIteration(A.Renderer(make_cfg(32, 16), None, sc), dec_b, gb, masks, tsdf, tb, STAGE_LR, use_graph=False)
    it_b.gstate = {k: (m.clone(), v.clone()) for k, (m, v) in it.gstate.items()}
    it_b.fstate = {k: (m.clone(), v.clone()) for k, (m, v) in it.fstate.items()}
    it_b.step_count.copy_(it.step_count)
    la, lb = float(it.step(*rays, 'color')), float(it_b.step(*rays, 'color'))
    assert abs(la - lb) <= 1e-6 * abs(lb), (la, lb)
    del junk


def test_bucket_has_a_slot_for_every_trainable_network():
    """fix_high: False (configs: src/Mapper.py:364-371 puts the high decoder into the optimiser): its flat gradient must live in
    the contiguous bucket like the others, or the dense all-reduce of the bucket prefix would leave it rank-local."""
    sc, dec, rend, rays, masks = setup()
    for p in dec.high_decoder.parameters():
        p.requires_grad_(True)
    grids = {k: v.clone().to(DEV) for k, v in sc.c.items()}
    it = mapping.MapperIteration(rend, dec, grids, None, sc.tsdf_volume.to(DEV), sc.tsdf_bnds.to(DEV), STAGE_LR, train=('high', 'color', 'att'),
                                 use_graph=False, distributed=False)
    names = [(kind, name) for kind, name, *_ in it._bucket_layout]
    assert ('flat', 'high') in names and ('flat', 'color') in names and ('flat', 'att') in names
    grids_g, flats_g = it._sequence(*[t.float().contiguous() for t in rays], 'color', False, adam=False)
    lo, hi = it.bucket.data_ptr(), it.bucket.data_ptr() + it.bucket.numel() * 4
    for n, f in flats_g.items():
        assert lo <= f.data_ptr() < hi, f'flat gradient of {n} is not a view of the bucket'
        assert float(f.abs().max()) > 0
    assert it.bucket_bytes == it.bucket.numel() * 4             # stage color produces the whole bucket: all of it is reduced
