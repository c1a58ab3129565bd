"""GPU (MI355X): the HIP path, called through the C ABI of libadfp.so by the reference-shaped
Python objects, against (1) the golden vectors generated from the reference and (2) the oracle
on seeded inputs.  Tolerance: BASELINE.json north_star -- <= 1e-4 relative, fp32."""
import numpy as np
import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import common, synthetic
from oracle import adfp_oracle as O
from conftest import make_cfg, to_dev, assert_close, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = 'cuda:0'


def build(mini_like, sd, n_samples=32, n_surface=16, **kw):
    dec = A.DF()
    dec.load_state_dict(sd)
    dec.bound = mini_like.bound
    dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(n_samples, n_surface, **kw), None, mini_like)
    return dec, rend


@pytest.fixture(scope='module')
def gm(mini):
    """mini scene on the GPU (the TSDF stays the permuted, non-contiguous view)."""
    class G(object):
        pass
    g = G()
    g.c = to_dev(mini.c, DEV)
    g.tsdf = mini.tsdf_volume.to(DEV)
    assert not g.tsdf.is_contiguous() and g.tsdf.stride() == mini.tsdf_volume.stride()
    g.tsdf_bnds = mini.tsdf_bnds.to(DEV)
    g.rays_o, g.rays_d = mini.rays_o.to(DEV), mini.rays_d.to(DEV)
    g.gt_depth, g.gt_color = mini.gt_depth.to(DEV), mini.gt_color.to(DEV)
    g.dec, g.rend = build(mini, mini.sd, mini.n_samples, mini.n_surface)
    return g


def test_native_library_is_loaded(gm):
    from attentive_dfprior_amd import _lib
    assert _lib.lib().adfp_version() == _lib.ABI_VERSION
    assert 'libadfp.so' in open('/proc/self/maps').read()


# --------------------------------------------------------------------------- golden vectors
@pytest.mark.parametrize('stage', O.STAGES)
def test_render_batch_ray_vs_reference_golden(mini, gm, stage):
    g = mini.golden(stage)
    with torch.no_grad():
        d, u, c, w = gm.rend.render_batch_ray(gm.c, gm.dec, gm.rays_d, gm.rays_o, DEV, gm.tsdf, gm.tsdf_bnds, stage,
                                              gt_depth=gm.gt_depth)
    assert d.dtype == torch.float64 and u.dtype == torch.float64           # Renderer.py:183-190
    assert c.dtype == torch.float32 and w.dtype == torch.float32
    assert tuple(w.shape) == g['weight'].shape
    assert_close(d, g['depth'], TOL, f'{stage} depth')
    assert_close(c, g['color'], TOL, f'{stage} color') if stage == 'color' else None
    assert_close(w, g['weight'], TOL, f'{stage} weight')
    assert_close(u, g['uncertainty'], TOL, f'{stage} uncertainty')


@pytest.mark.parametrize('stage', O.STAGES)
def test_intermediates_vs_golden(mini, gm, stage):
    """z_vals bit-exact; raw within tolerance; report band-mask flips (discontinuity)."""
    g = mini.golden(stage)
    with torch.no_grad():
        d, u, c, w, aux = gm.rend._engine.render_forward(
            gm.dec, gm.c, gm.rays_o, gm.rays_d, gm.gt_depth, gm.tsdf, gm.tsdf_bnds, mini.bound, stage,
            mini.n_samples, mini.n_surface, want_aux=True)
    z = aux['z_vals'].cpu().numpy()
    assert np.abs(z - g['z_vals']).max() <= 4e-16 * np.abs(g['z_vals']).max()
    assert (np.diff(z, axis=1) >= 0).all()
    raw = aux['raw'].cpu()
    ref = torch.from_numpy(g['raw'])
    assert torch.equal(raw[..., 3] == 100, ref[..., 3] == 100), 'out-of-bound sample sets differ'
    flips = int(((w.cpu().reshape(-1) == 1) != (torch.from_numpy(g['weight']).reshape(-1) == 1)).sum())
    assert flips == 0, f'{flips} samples flipped across the TSDF band mask'
    assert_close(raw[..., 3], ref[..., 3], TOL, f'{stage} occ')
    if stage == 'color':
        assert_close(raw[..., :3], ref[..., :3], TOL, 'rgb')


@pytest.mark.parametrize('stage', O.STAGES)
def test_render_without_sensor_depth_vs_golden(mini, gm, stage):
    g = mini.golden(stage)
    with torch.no_grad():
        d, u, c, w = gm.rend.render_batch_ray(gm.c, gm.dec, gm.rays_d, gm.rays_o, DEV, gm.tsdf, gm.tsdf_bnds, stage)
    assert tuple(w.shape) == g['nd_weight'].shape
    assert_close(d, g['nd_depth'], TOL, 'nd depth')
    assert_close(c, g['nd_color'], TOL, 'nd color')
    assert_close(w, g['nd_weight'], TOL, 'nd weight')


@pytest.mark.parametrize('stage', O.STAGES)
def test_eval_points_and_df_forward_vs_golden(mini, gm, stage):
    g = mini.golden(stage)
    qp = mini.query_points.to(DEV)
    with torch.no_grad():
        raw, w = gm.rend.eval_points(qp, gm.dec, gm.tsdf, gm.tsdf_bnds, gm.c, stage, DEV)
        raw2, w2 = gm.dec(qp.unsqueeze(0), c_grid=gm.c, tsdf_volume=gm.tsdf, tsdf_bnds=gm.tsdf_bnds, stage=stage)
        raw3, _ = gm.rend.eval_points(qp.float(), gm.dec, gm.tsdf, gm.tsdf_bnds, gm.c, stage, DEV)
    assert torch.equal(raw[:, 3].cpu() == 100, torch.from_numpy(g['q_raw'])[:, 3] == 100)
    assert_close(raw, g['q_raw'], TOL, 'eval_points raw')
    assert_close(w, g['q_w'], TOL, 'eval_points w')
    assert_close(raw2, g['df_raw'], TOL, 'DF.forward raw (no bound rule)')
    assert_close(w2, g['df_w'], TOL, 'DF.forward w')
    assert raw3.shape == raw.shape


def test_rays_tsdf_image_vs_golden(mini, gm):
    g = mini.golden('rays')
    ro, rd = common.get_rays(mini.H, mini.W, mini.fx, mini.fy, mini.cx, mini.cy, mini.c2w.to(DEV), DEV)
    assert np.array_equal(ro.cpu().numpy(), g['get_rays_o'])
    assert np.abs(rd.cpu().numpy() - g['get_rays_d']).max() <= 1.2e-7 * np.abs(g['get_rays_d']).max()
    t = gm.rend.eval_points_tsdf(mini.query_points.to(DEV), gm.tsdf, DEV)
    assert tuple(t.shape) == g['tsdf_q'].shape
    assert np.abs(t.cpu().numpy() - g['tsdf_q']).max() <= 5e-7
    gm.rend.ray_batch_size = int(g['img_ray_batch_size'])
    try:
        di, ui, ci = gm.rend.render_img(gm.c, gm.dec, mini.c2w.to(DEV), DEV, gm.tsdf, gm.tsdf_bnds, 'color',
                                        gt_depth=mini.depth_img.to(DEV))
    finally:
        gm.rend.ray_batch_size = 100000
    assert di.dtype == torch.float64 and tuple(di.shape) == (mini.H, mini.W) and tuple(ci.shape) == (mini.H, mini.W, 3)
    assert_close(di, g['img_depth'], TOL, 'img depth')
    assert_close(ci, g['img_color'], TOL, 'img color')
    mse = float(((ci.cpu().double() - torch.from_numpy(g['img_color']).double()) ** 2).mean())
    rng = float(np.abs(g['img_color']).max())
    psnr = 10 * np.log10(rng * rng / max(mse, 1e-30))
    assert psnr > 80.0, psnr


# --------------------------------------------------------------------------- oracle, seeded
@pytest.mark.parametrize('n_samples,n_surface', [(48, 16), (96, 32), (17, 5), (64, 0), (1, 1)])
@pytest.mark.parametrize('stage', ['low', 'color'])
def test_sample_counts_vs_oracle(mini, gm, stage, n_samples, n_surface):
    """S = 64 (benchmark), 128 (config 5), ragged, no surface samples, degenerate."""
    dec, rend = build(mini, mini.sd, n_samples, n_surface)
    with torch.no_grad():
        d, u, c, w = rend.render_batch_ray(gm.c, dec, gm.rays_d, gm.rays_o, DEV, gm.tsdf, gm.tsdf_bnds, stage,
                                           gt_depth=gm.gt_depth)
    od, ou, oc, ow = O.render_batch_ray(mini.sd, mini.c, mini.rays_d, mini.rays_o, mini.tsdf_volume, mini.tsdf_bnds,
                                        mini.bound, stage, mini.gt_depth, n_samples, n_surface)
    assert tuple(w.shape) == tuple(ow.shape)
    assert_close(d, od, TOL, 'depth')
    assert_close(c, oc, TOL, 'color')
    assert_close(w, ow, TOL, 'weight')


@pytest.mark.parametrize('kw', [{'perturb': 1.0}, {'lindisp': True}])
def test_sampler_variants_vs_oracle(mini, gm, kw):
    dec, rend = build(mini, mini.sd, 32, 16, **kw)
    keep = mini.gt_depth > 0 if kw.get('lindisp') else torch.ones_like(mini.gt_depth, dtype=torch.bool)
    ro, rd, gd = mini.rays_o[keep], mini.rays_d[keep], mini.gt_depth[keep]
    torch.manual_seed(21)
    with torch.no_grad():
        d, u, c, w = rend.render_batch_ray(gm.c, dec, rd.to(DEV), ro.to(DEV), DEV, gm.tsdf, gm.tsdf_bnds, 'color',
                                           gt_depth=gd.to(DEV))
    torch.manual_seed(21)
    t_rand = torch.rand(ro.shape[0], 32) if kw.get('perturb') else None
    od, ou, oc, ow = O.render_batch_ray(mini.sd, mini.c, rd, ro, mini.tsdf_volume, mini.tsdf_bnds, mini.bound,
                                        'color', gd, 32, 16, lindisp=kw.get('lindisp', False),
                                        perturb=kw.get('perturb', 0.0), t_rand=t_rand)
    assert_close(d, od, TOL, 'depth')
    assert_close(c, oc, TOL, 'color')


def test_second_seed_and_larger_scene_vs_oracle():
    """A different weight seed on the 'tiny' scene (2.8 m room, 1/25 m TSDF, default-init grids)."""
    sc = synthetic.Scene('tiny', H=60, W=80, fx=72.2, fy=72.2, cx=39.5, cy=29.5, voxel=0.04, grid_std_scale=20.0,
                         seed=7)
    sc.c['grid_high'] = sc.c['grid_high'] * 100
    sd = O.random_state_dict(seed=11)
    ro, rd, gd, col = synthetic.make_ray_batch(sc, 600, seed=2, poses=3)
    dec, rend = build(sc, sd, 48, 16)
    with torch.no_grad():
        d, u, c, w = rend.render_batch_ray(to_dev(sc.c, DEV), dec, rd.to(DEV), ro.to(DEV), DEV,
                                           sc.tsdf_volume.to(DEV), sc.tsdf_bnds.to(DEV), 'color', gt_depth=gd.to(DEV))
    od, ou, oc, ow = O.render_batch_ray(sd, sc.c, rd, ro, sc.tsdf_volume, sc.tsdf_bnds, sc.bound, 'color', gd, 48, 16)
    flips = int(((w.cpu() == 1) != (ow == 1)).sum())
    assert flips <= 2, flips
    assert_close(d, od, TOL, 'depth')
    assert_close(c, oc, TOL, 'color')
    assert_close(u, ou, TOL, 'uncertainty')


def test_composite_entry_vs_oracle(mini):
    g = torch.Generator().manual_seed(0)
    raw = torch.randn(333, 70, 4, generator=g) * 0.3
    raw[5, :, 3] = 100.0
    raw[6, :, 3] = -100.0
    z = torch.sort(torch.rand(333, 70, generator=g, dtype=torch.float64) * 5, dim=1)[0]
    d, v, rgb, wts = common.raw2outputs_nerf_color(raw.to(DEV), z.to(DEV), None, occupancy=True, device=DEV)
    od, ov, orgb, ow = O.raw2outputs(raw.clone(), z)
    # sums of 70 signed terms: every element within 2e-6 of the tensor's scale (f64 depth: 2e-7), then the north-star bar per element
    from conftest import assert_close_scale
    assert_close_scale(d, od, 2e-7, 'depth')
    assert_close_scale(rgb, orgb, 2e-6, 'rgb')
    assert_close_scale(wts, ow, 2e-6, 'weights')
    assert_close_scale(v, ov, 2e-6, 'var')
    for got, ref, what in ((d, od, 'depth'), (rgb, orgb, 'rgb'), (wts, ow, 'weights'), (v, ov, 'var')):
        assert_close(got, ref, 1e-4, what)


def test_relayout_round_trip(gm):
    from attentive_dfprior_amd import _lib
    L = _lib.lib()
    g = torch.randn(1, 32, 7, 9, 11, device=DEV)
    cl = torch.empty(7, 9, 11, 32, device=DEV)
    back = torch.empty_like(g)
    st = _lib.current_stream(torch.device(DEV))
    assert L.adfp_relayout_grid(_lib.ptr(g), _lib.ptr(cl), 32, 7, 9, 11, st) == 0
    assert L.adfp_relayout_grid_back(_lib.ptr(cl), _lib.ptr(back), 32, 7, 9, 11, st) == 0
    assert torch.equal(cl, g[0].permute(1, 2, 3, 0).contiguous())
    assert torch.equal(back, g)


def test_relayout_several_grids_in_one_launch(mini, gm):
    """adfp_relayout_grids (round 6): up to four grids of different sizes in one launch, both directions, against plain permutes;
    and the form the render path uses -- the conversions handed to the render call's FIRST launch (adfp_render_args.relayout_jobs)
    give the outputs of a call that found the channels-last copies ready, bit for bit."""
    from attentive_dfprior_amd import _lib
    L = _lib.lib()
    st = _lib.current_stream(torch.device(DEV))
    shapes = [(7, 9, 11), (1, 1, 1), (13, 5, 64), (3, 70, 2)]
    gs = [torch.randn(1, 32, *s, device=DEV) for s in shapes]
    cls = [torch.empty(*s, 32, device=DEV) for s in shapes]
    arr = (_lib.AdfpRelayoutJob * 4)()
    for k in range(4):
        arr[k].src, arr[k].dst, arr[k].voxels = gs[k].data_ptr(), cls[k].data_ptr(), shapes[k][0] * shapes[k][1] * shapes[k][2]
    assert L.adfp_relayout_grids(4, arr, 0, st) == 0
    for g, cl in zip(gs, cls):
        assert torch.equal(cl, g[0].permute(1, 2, 3, 0).contiguous())
    backs = [torch.empty_like(g) for g in gs]
    for k in range(4):
        arr[k].src, arr[k].dst = cls[k].data_ptr(), backs[k].data_ptr()
    assert L.adfp_relayout_grids(3, arr, 1, st) == 0                      # three of the four
    for k in range(3):
        assert torch.equal(backs[k], gs[k])
    assert L.adfp_relayout_grids(5, arr, 0, st) < 0 and L.adfp_relayout_grids(0, None, 0, st) == 0
    # through the render call: cold caches (the first launch converts) against warm ones
    eng = gm.rend._engine
    with torch.no_grad():
        eng._grid_cache.clear()
        cold = gm.rend.render_batch_ray(gm.c, gm.dec, gm.rays_d, gm.rays_o, DEV, gm.tsdf, gm.tsdf_bnds, 'color', gt_depth=gm.gt_depth)
        assert len(eng._grid_cache) == 3
        warm = gm.rend.render_batch_ray(gm.c, gm.dec, gm.rays_d, gm.rays_o, DEV, gm.tsdf, gm.tsdf_bnds, 'color', gt_depth=gm.gt_depth)
    for x, y in zip(cold, warm):
        assert torch.equal(x, y)
    for key in ('grid_low', 'grid_high', 'grid_color'):
        assert torch.equal(eng._grid_cache[key][1], gm.c[key][0].permute(1, 2, 3, 0).contiguous())


# --------------------------------------------------------------------------- edge cases
def test_empty_and_single_ray(mini, gm):
    with torch.no_grad():
        d, u, c, w = gm.rend.render_batch_ray(gm.c, gm.dec, gm.rays_d[:0], gm.rays_o[:0], DEV, gm.tsdf, gm.tsdf_bnds,
                                              'color', gt_depth=gm.gt_depth[:0])
        assert d.shape == (0,) and c.shape == (0, 3) and w.shape == (0, 48, 1)
        d, u, c, w = gm.rend.render_batch_ray(gm.c, gm.dec, gm.rays_d[:1], gm.rays_o[:1], DEV, gm.tsdf, gm.tsdf_bnds,
                                              'color', gt_depth=gm.gt_depth[:1])
    od, ou, oc, ow = O.render_batch_ray(mini.sd, mini.c, mini.rays_d[:1], mini.rays_o[:1], mini.tsdf_volume,
                                        mini.tsdf_bnds, mini.bound, 'color', mini.gt_depth[:1], 32, 16)
    assert_close(d, od, TOL, 'depth')
    assert_close(c, oc, TOL, 'color')
    raw, w = gm.rend.eval_points(mini.query_points[:0].to(DEV), gm.dec, gm.tsdf, gm.tsdf_bnds, gm.c, 'color', DEV)
    assert raw.shape == (0, 4)


def test_all_zero_depth_and_rays_leaving_bound(mini, gm):
    """Zero-depth rays sample 0.001..max(gt_depth) (Renderer.py:191-201); rays that start outside
    the bound get occ = 100 on every sample."""
    n = 40
    ro = mini.rays_o[:n].clone()
    ro[n // 2:] += torch.tensor([5.0, 0.0, 0.0])          # outside the scene bound
    gd = torch.zeros(n)
    gd[0] = 0.7
    with torch.no_grad():
        d, u, c, w = gm.rend.render_batch_ray(gm.c, gm.dec, gm.rays_d[:n], ro.to(DEV), DEV, gm.tsdf, gm.tsdf_bnds,
                                              'color', gt_depth=gd.to(DEV))
    od, ou, oc, ow = O.render_batch_ray(mini.sd, mini.c, mini.rays_d[:n], ro, mini.tsdf_volume, mini.tsdf_bnds,
                                        mini.bound, 'color', gd, 32, 16)
    assert torch.isfinite(d).all() and torch.isfinite(c).all()
    assert_close(d, od, TOL, 'depth')
    assert_close(c, oc, TOL, 'color')
    assert_close(w, ow, TOL, 'weight')


def test_deterministic_and_order_invariant(gm):
    """Bitwise reproducible across launches (the in-band list order is not, the outputs are) and
    independent of how rays are tiled: a permuted batch gives the permuted outputs bit for bit."""
    with torch.no_grad():
        a = gm.rend.render_batch_ray(gm.c, gm.dec, gm.rays_d, gm.rays_o, DEV, gm.tsdf, gm.tsdf_bnds, 'color',
                                     gt_depth=gm.gt_depth)
        b = gm.rend.render_batch_ray(gm.c, gm.dec, gm.rays_d, gm.rays_o, DEV, gm.tsdf, gm.tsdf_bnds, 'color',
                                     gt_depth=gm.gt_depth)
        perm = torch.randperm(gm.rays_o.shape[0], device=DEV)
        p = gm.rend.render_batch_ray(gm.c, gm.dec, gm.rays_d[perm], gm.rays_o[perm], DEV, gm.tsdf, gm.tsdf_bnds,
                                     'color', gt_depth=gm.gt_depth[perm])
    for x, y, z in zip(a, b, p):
        assert torch.equal(x, y)
        assert torch.equal(x[perm], z)


def test_sharded_render_equals_whole(gm):
    """Two ray shards given the full-batch depth max reproduce the unsharded render bit for bit
    (what attentive_dfprior_amd.dist relies on); without it the far clamp differs."""
    n = gm.rays_o.shape[0]
    h = n // 3
    with torch.no_grad():
        whole = gm.rend.render_batch_ray(gm.c, gm.dec, gm.rays_d, gm.rays_o, DEV, gm.tsdf, gm.tsdf_bnds, 'color',
                                         gt_depth=gm.gt_depth)
        dmax = gm.gt_depth.max().reshape(1)
        parts = [gm.rend.render_batch_ray(gm.c, gm.dec, gm.rays_d[s], gm.rays_o[s], DEV, gm.tsdf, gm.tsdf_bnds,
                                          'color', gt_depth=gm.gt_depth[s], depth_max=dmax)
                 for s in (slice(0, h), slice(h, n))]
    for k in range(4):
        assert torch.equal(whole[k], torch.cat([parts[0][k], parts[1][k]], dim=0))


def test_weight_update_invalidates_packed_cache(mini, gm):
    dec, rend = build(mini, mini.sd)
    with torch.no_grad():
        a = rend.render_batch_ray(gm.c, dec, gm.rays_d, gm.rays_o, DEV, gm.tsdf, gm.tsdf_bnds, 'color', gt_depth=gm.gt_depth)
        dec.color_decoder.output_linear.bias.add_(0.25)       # in-place, like optimizer.step()
        b = rend.render_batch_ray(gm.c, dec, gm.rays_d, gm.rays_o, DEV, gm.tsdf, gm.tsdf_bnds, 'color', gt_depth=gm.gt_depth)
        c2 = {k: v.clone() for k, v in gm.c.items()}
        c2['grid_color'] += 0.05
        c3 = rend.render_batch_ray(c2, dec, gm.rays_d, gm.rays_o, DEV, gm.tsdf, gm.tsdf_bnds, 'color', gt_depth=gm.gt_depth)
    assert torch.equal(a[0], b[0]) and not torch.equal(a[2], b[2])
    assert not torch.equal(b[2], c3[2])


# --------------------------------------------------------------------------- full size (config 2)
def test_full_frame_room0_properties():
    """640x480, 64 samples/ray on the room0-sized scene: size-independent properties + a ray subset
    against the oracle."""
    sc = synthetic.Scene('room0', device=DEV, grid_std_scale=20.0)
    sc.c['grid_high'] = sc.c['grid_high'] * 100
    sd = O.random_state_dict(seed=0)
    dec, rend = build(sc, sd, 48, 16)
    c2w = sc.default_c2w()
    gd = sc.depth_image(c2w)
    di, ui, ci = rend.render_img(sc.c, dec, c2w, DEV, sc.tsdf_volume, sc.tsdf_bnds, 'color', gt_depth=gd)
    assert torch.isfinite(di).all() and torch.isfinite(ci).all() and (ui >= -1e-9).all()
    assert (di >= 0).all() and float(di.max()) <= 1.2 * float(gd.max()) + 0.02
    # subset of rays of the first render_img batch through the oracle (same per-batch depth max)
    ro, rd = common.get_rays(sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, c2w, DEV)
    ro, rd, g = ro.reshape(-1, 3)[:100000], rd.reshape(-1, 3)[:100000], gd.reshape(-1)[:100000]
    pick = torch.arange(0, 100000, 997, device=DEV)
    pick[0] = int(torch.argmax(g))                       # keep the batch max so the far clamp is identical
    cpu = {k: v.cpu() for k, v in sc.c.items()}
    od, ou, oc, ow = O.render_batch_ray(sd, cpu, rd[pick].cpu(), ro[pick].cpu(), sc.tsdf_volume.cpu(),
                                        sc.tsdf_bnds, sc.bound, 'color', g[pick].cpu(), 48, 16)
    assert_close(di.reshape(-1)[pick], od, TOL, 'depth')
    assert_close(ci.reshape(-1, 3)[pick], oc, TOL, 'color')


def test_tsdf_generic_strides_path(mini, gm):
    """A contiguous [1,1,Z,Y,X] TSDF (X fastest) takes the strided 8-load path; the permuted view of
    the reference (Z fastest) takes the paired-load path.  Same values either way."""
    qp = mini.query_points.to(DEV)
    a = gm.rend.eval_points_tsdf(qp, gm.tsdf, DEV)
    b = gm.rend.eval_points_tsdf(qp, gm.tsdf.contiguous(), DEV)
    assert torch.equal(a, b)
    # top z boundary (z0 = Z-1) and beyond: border clamp
    lo, hi = mini.tsdf_bnds[:, 0], mini.tsdf_bnds[:, 1]
    edge = torch.stack([lo + (hi - lo) * torch.tensor([0.3, 0.6, 1.0], dtype=torch.float64),
                        lo + (hi - lo) * torch.tensor([0.5, 0.5, 1.2], dtype=torch.float64),
                        lo + (hi - lo) * torch.tensor([1.0, 1.0, 1.0], dtype=torch.float64),
                        lo + (hi - lo) * torch.tensor([0.0, 0.0, 0.0], dtype=torch.float64)])
    got = gm.rend.eval_points_tsdf(edge.to(DEV), gm.tsdf, DEV).cpu().reshape(-1)
    ref = O.trilerp(mini.tsdf_volume, edge, mini.tsdf_bnds).reshape(-1)
    assert (got - ref).abs().max().item() <= 5e-7


@pytest.mark.parametrize('n', [0, 1, 63, 1023, 1024, 1025, 5000, 20001])
def test_prefilter_rays_vs_oracle(mini, n):
    """a3 (src/Mapper.py:438-449): kept rays and their ORDER equal boolean-mask indexing; zero direction
    components (inf / NaN in the slab test) and exact-equality depths included; autograd survives."""
    dev = torch.device('cuda:0')
    scene = synthetic.mini_scene()
    ro, rd, depth, color = synthetic.make_ray_batch(scene, max(n, 1), seed=11 + n, zero_frac=0.2)
    ro, rd, depth, color = ro[:n], rd[:n], depth[:n], color[:n]
    g = torch.Generator().manual_seed(n)
    depth = depth * (0.5 + 1.5 * torch.rand(depth.shape, generator=g))      # some beyond the box -> dropped
    if n >= 63:
        rd[3, 0] = 0.0                          # +-inf on one axis
        rd[5] = 0.0                             # inf everywhere
        ro[7, 1] = float(scene.bound[1, 0]); rd[7, 1] = 0.0      # 0/0 = NaN on one plane -> ray dropped
        t = (scene.bound.unsqueeze(0) - ro[9:10].unsqueeze(-1)) / rd[9:10].unsqueeze(-1)
        depth[9] = torch.min(torch.max(t, dim=2)[0], dim=1)[0].float()      # t == depth after f32 rounding or not
    mask = O.prefilter_mask(ro, rd, depth, scene.bound)
    ro_g = ro.to(dev).requires_grad_(True)
    o, d, z, c = common.filter_rays_in_bound(ro_g, rd.to(dev), depth.to(dev), color.to(dev), scene.bound)
    assert o.shape[0] == int(mask.sum())
    assert torch.equal(o.detach().cpu(), ro[mask]) and torch.equal(d.cpu(), rd[mask])
    assert torch.equal(z.cpu(), depth[mask]) and torch.equal(c.cpu(), color[mask])
    if n:
        o.sum().backward()
        assert torch.equal(ro_g.grad.cpu(), mask.float().unsqueeze(-1).expand(-1, 3))


def test_prefilter_rays_vs_reference_lines():
    """a3 against the vectors the reference's own lines (src/Mapper.py:438-449) produced, executed by
    tests/golden/make_mapper_golden.py: kept set, order, and the kept rows themselves, bit for bit."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, 'mapper_prefilter.npz'))
    names = sorted({k.split('.')[0] for k in g.files if '.' in k})
    for n in names:
        ro, rd, gd, gc = (torch.from_numpy(g[f'{n}.{k}']).to(DEV) for k in ('rays_o', 'rays_d', 'gt_depth', 'gt_color'))
        o, d, z, c = common.filter_rays_in_bound(ro, rd, gd, gc, torch.from_numpy(g[f'{n}.bound']))
        mask = torch.from_numpy(g[f'{n}.inside_mask'])
        assert o.shape[0] == int(mask.sum()), n
        assert np.array_equal(o.cpu().numpy(), g[f'{n}.kept_rays_o'], equal_nan=True), n
        assert np.array_equal(z.cpu().numpy(), g[f'{n}.kept_gt_depth'], equal_nan=True), n
        assert torch.equal(c.cpu(), torch.from_numpy(g[f'{n}.gt_color'])[mask]), n


# --------------------------------------------------------------------------- a2: rays through given pixels
def test_get_rays_from_uv_vs_golden_and_pose_gradient(mini):
    """a2 (src/common.py:76-91) on the device: the reference's own vectors, then a larger seeded batch against the oracle
    (values) and against torch autograd of the oracle's formula (gradient w.r.t. the camera pose)."""
    g = mini.golden('rays')
    i, j = torch.from_numpy(g['uv_i']).to(DEV), torch.from_numpy(g['uv_j']).to(DEV)
    ro, rd = common.get_rays_from_uv(i, j, mini.c2w.to(DEV), mini.H, mini.W, mini.fx, mini.fy, mini.cx, mini.cy, DEV)
    assert np.array_equal(ro.cpu().numpy(), g['uv_rays_o'])
    assert np.abs(rd.cpu().numpy() - g['uv_rays_d']).max() <= 1.2e-7 * np.abs(g['uv_rays_d']).max()
    gen = torch.Generator().manual_seed(4)
    n = 1537
    i = torch.rand(n, generator=gen) * (mini.W - 1)
    j = torch.rand(n, generator=gen) * (mini.H - 1)
    c2w = mini.c2w.clone().requires_grad_(True)
    oro, ord_ = O.get_rays_from_uv(i, j, c2w, mini.fx, mini.fy, mini.cx, mini.cy)
    wo, wd = torch.randn(n, 3, generator=gen), torch.randn(n, 3, generator=gen)
    ((oro * wo).sum() + (ord_ * wd).sum()).backward()
    c2w_g = mini.c2w.clone().to(DEV).requires_grad_(True)
    ro, rd = common.get_rays_from_uv(i.to(DEV), j.to(DEV), c2w_g, mini.H, mini.W, mini.fx, mini.fy, mini.cx, mini.cy, DEV)
    assert torch.equal(ro.detach().cpu(), oro.detach()) and (rd.detach().cpu() - ord_.detach()).abs().max() <= 2e-7 * ord_.abs().max()
    ((ro * wo.to(DEV)).sum() + (rd * wd.to(DEV)).sum()).backward()
    from conftest import assert_close_scale
    assert_close_scale(c2w_g.grad, c2w.grad, 1e-5, 'd/d c2w')          # a sum over the rays
    assert_close(c2w_g.grad[:3], c2w.grad[:3], 1e-4, 'd/d c2w, per element')      # (the bottom row of c2w takes no part in a ray: 0 on both sides)
    assert 'libadfp.so' in open('/proc/self/maps').read()


def test_get_samples_rng_stream_is_torch_randint(mini):
    """a2: pixel selection keeps the reference's RNG -- ONE torch.randint(H*W, (n,)) draw on the device per call
    (src/common.py:101) -- so a seeded run selects the reference's pixels; depth / colour follow the same indices."""
    H, W = mini.H, mini.W
    depth = mini.depth_img.to(DEV)
    color = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(2)).to(DEV)
    torch.manual_seed(123)
    ro, rd, d, c = common.get_samples(0, H, 0, W, 77, H, W, mini.fx, mini.fy, mini.cx, mini.cy, mini.c2w.to(DEV), depth, color, DEV)
    torch.manual_seed(123)
    idx = torch.randint(H * W, (77,), device=DEV)
    assert torch.equal(d, depth.reshape(-1)[idx]) and torch.equal(c, color.reshape(-1, 3)[idx])
    jj, ii = idx // W, idx % W
    oro, ord_ = O.get_rays_from_uv(ii.float().cpu(), jj.float().cpu(), mini.c2w, mini.fx, mini.fy, mini.cx, mini.cy)
    assert (rd.cpu() - ord_).abs().max() <= 2e-7 * ord_.abs().max() and torch.equal(ro.cpu(), oro)


def test_render_img_single_call_equals_the_batched_loop_bit_for_bit(mini, gm):
    """render_img renders the frame as ONE call with a max(gt_depth) per ray_batch_size segment (adfp_render_args.depth_max_segment);
    the reference's loop gives every batch its own maximum (src/utils/Renderer.py:294-313).  Same values, bit for bit -- also
    when the last segment is ragged and when a segment's maximum is far below the frame's."""
    from attentive_dfprior_amd.common import get_rays
    dimg = mini.depth_img.to(DEV).clone()
    dimg[: mini.H // 3] *= 0.35                                   # the first segments see a much smaller maximum than the rest
    c2w = mini.c2w.to(DEV)
    for bs in (1000, 700):                                        # 3 072 rays: 4 segments (72-ray tail) / 5 segments (272-ray tail)
        rend = A.Renderer(make_cfg(mini.n_samples, mini.n_surface), None, mini, ray_batch_size=bs)
        with torch.no_grad():
            d1, u1, c1 = rend.render_img(gm.c, gm.dec, c2w, DEV, gm.tsdf, gm.tsdf_bnds, 'color', gt_depth=dimg)
            ro, rd = get_rays(mini.H, mini.W, mini.fx, mini.fy, mini.cx, mini.cy, c2w, DEV)
            ro, rd, gd = ro.reshape(-1, 3), rd.reshape(-1, 3), dimg.reshape(-1)
            parts = [rend.render_batch_ray(gm.c, gm.dec, rd[i:i + bs], ro[i:i + bs], DEV, gm.tsdf, gm.tsdf_bnds, 'color', gt_depth=gd[i:i + bs])
                     for i in range(0, ro.shape[0], bs)]
        assert torch.equal(d1.reshape(-1), torch.cat([p[0] for p in parts]))
        assert torch.equal(u1.reshape(-1), torch.cat([p[1] for p in parts]))
        assert torch.equal(c1.reshape(-1, 3), torch.cat([p[2] for p in parts]))
        # and the segment maxima matter: one maximum for the whole frame gives other values
        with torch.no_grad():
            whole = rend.render_batch_ray(gm.c, gm.dec, rd, ro, DEV, gm.tsdf, gm.tsdf_bnds, 'color', gt_depth=gd)
        assert not torch.equal(whole[0], d1.reshape(-1))


def test_render_img_shards_concatenate_to_render_img_bit_for_bit(mini, gm):
    """Renderer.render_img_shard (one GPU's contiguous pixel range of a ray-sharded frame, dist.render_img_sharded): shards whose
    borders cut across the ray_batch_size segments, ragged and empty shards included, concatenate to the frame render_img returns
    -- the far clamps come from the whole frame's segment maxima (adfp_render_args.depth_max_first_ray)."""
    from attentive_dfprior_amd import dist as adist
    dimg = mini.depth_img.to(DEV).clone()
    dimg[: mini.H // 3] *= 0.35
    c2w = mini.c2w.to(DEV)
    n = mini.H * mini.W
    for bs, world in ((1000, 8), (700, 5), (1000, 3)):
        rend = A.Renderer(make_cfg(mini.n_samples, mini.n_surface), None, mini, ray_batch_size=bs)
        d1, u1, c1 = rend.render_img(gm.c, gm.dec, c2w, DEV, gm.tsdf, gm.tsdf_bnds, 'color', gt_depth=dimg)
        assert torch.equal(rend.segment_depth_max(dimg), torch.stack([dimg.reshape(-1)[i:i + bs].max() for i in range(0, n, bs)]))
        parts = [rend.render_img_shard(gm.c, gm.dec, c2w, DEV, gm.tsdf, gm.tsdf_bnds, 'color', dimg, *adist.shard_range(n, r, world))
                 for r in range(world)]
        assert torch.equal(torch.cat([p[0] for p in parts]), d1.reshape(-1)), (bs, world)
        assert torch.equal(torch.cat([p[1] for p in parts]), u1.reshape(-1)), (bs, world)
        assert torch.equal(torch.cat([p[2] for p in parts]), c1.reshape(-1, 3)), (bs, world)
    empty = rend.render_img_shard(gm.c, gm.dec, c2w, DEV, gm.tsdf, gm.tsdf_bnds, 'color', dimg, 17, 17)
    assert empty[0].shape == (0,) and empty[2].shape == (0, 3)
    whole = adist.render_img_sharded(rend, gm.c, gm.dec, c2w, DEV, gm.tsdf, gm.tsdf_bnds, 'color', dimg)      # no process group: one rank
    assert torch.equal(whole[0], d1) and torch.equal(whole[2], c1)


def test_frame_job_equals_explicit_rays_and_explicit_maxima_bit_for_bit(mini, gm):
    """adfp_frame_job (round 6): the call's first launch writes the rays of its own pixels and reduces the WHOLE frame's per-segment
    max(gt_depth) as 16 partial maxima per segment that the sampler's lanes fold.  Against the explicit form of the same call --
    common.get_rays' rays of the pixel range, the maxima computed by torch (Renderer.segment_depth_max) and handed in as depth_max --
    bit for bit, for pixel ranges that start and end inside segments, a ragged last segment, and a segment of invalid (zero) depth."""
    from attentive_dfprior_amd.common import get_rays
    dimg = mini.depth_img.to(DEV).clone()
    dimg[: mini.H // 3] *= 0.35
    dimg[mini.H // 3: mini.H // 3 + 12] = 0.0                     # a whole 700-ray segment without a valid depth: maximum 0
    c2w = mini.c2w.to(DEV)
    n = mini.H * mini.W
    ro, rd = get_rays(mini.H, mini.W, mini.fx, mini.fy, mini.cx, mini.cy, c2w, DEV)
    ro, rd, gd = ro.reshape(-1, 3), rd.reshape(-1, 3), dimg.reshape(-1)
    for bs in (700, 1000, 64):
        rend = A.Renderer(make_cfg(mini.n_samples, mini.n_surface), None, mini, ray_batch_size=bs)
        seg_max = rend.segment_depth_max(dimg)
        for lo, hi in ((0, n), (333, 2111), (n - 5, n), (1400, 1401)):
            with torch.no_grad():
                a = rend.render_img_shard(gm.c, gm.dec, c2w, DEV, gm.tsdf, gm.tsdf_bnds, 'color', dimg, lo, hi)
                b = rend._engine.render_forward(gm.dec, gm.c, ro[lo:hi], rd[lo:hi], gd[lo:hi], gm.tsdf, gm.tsdf_bnds, mini.bound, 'color',
                                                mini.n_samples, mini.n_surface, depth_max=seg_max, depth_max_segment=bs, depth_max_first_ray=lo)
            for x, y, what in zip(a, b, ('depth', 'uncertainty', 'colour')):
                assert torch.equal(x, y), (bs, lo, hi, what)
    # the unsegmented call (render_batch_ray without depth_max): one "segment" = the whole call, the same partial maxima
    with torch.no_grad():
        a = rend._engine.render_forward(gm.dec, gm.c, ro[100:1777], rd[100:1777], gd[100:1777], gm.tsdf, gm.tsdf_bnds, mini.bound, 'color',
                                        mini.n_samples, mini.n_surface)
        b = rend._engine.render_forward(gm.dec, gm.c, ro[100:1777], rd[100:1777], gd[100:1777], gm.tsdf, gm.tsdf_bnds, mini.bound, 'color',
                                        mini.n_samples, mini.n_surface, depth_max=gd[100:1777].max().reshape(1))
    for x, y in zip(a[:4], b[:4]):
        assert torch.equal(x, y)


def test_prefilter_job_equals_the_prefilter_launch_bit_for_bit(mini, gm):
    """adfp_render_args.prefilter_bound (round 6): the Mapper's bounding-box pre-filter (src/Mapper.py:438-449) as a job of the render
    call's first launch -- keep flags and the kept rays' maximum depth -- against adfp_prefilter_mask + depth_max: the same flags, the
    same outputs bit for bit, for a batch with dropped rays (depth beyond the bound), zero depths and a NaN direction; and a batch
    whose rays are ALL dropped (maximum = -inf on both sides)."""
    from attentive_dfprior_amd import _lib
    L = _lib.lib()
    eng = gm.rend._engine
    st = _lib.current_stream(torch.device(DEV))
    bound_dev = torch.as_tensor(mini.bound).to(DEV, torch.float64).contiguous()
    ro, rd, gd, _ = [t.to(DEV) for t in synthetic.make_ray_batch(synthetic.mini_scene(), 1500, seed=11, poses=3)]
    gd = gd.clone()
    gd[::7] *= 40.0                                               # beyond the bound: dropped
    rd = rd.clone()
    rd[5, 1] = float('nan')
    for case in ('mixed', 'all dropped'):
        if case == 'all dropped':
            gd = gd * 0 + 1e6
        keep_a = torch.empty((ro.shape[0],), dtype=torch.uint8, device=DEV)
        dmax = torch.empty((1,), dtype=torch.float32, device=DEV)
        _lib.check(L.adfp_prefilter_mask(_lib.ptr(ro), _lib.ptr(rd), _lib.ptr(gd), ro.shape[0], _lib.ptr(bound_dev), _lib.ptr(keep_a), _lib.ptr(dmax), st), 'prefilter')
        keep_b = torch.zeros_like(keep_a) + 7
        with torch.no_grad():
            a = eng.render_forward(gm.dec, gm.c, ro, rd, gd, gm.tsdf, gm.tsdf_bnds, mini.bound, 'color', mini.n_samples, mini.n_surface, depth_max=dmax)
            b = eng.render_forward(gm.dec, gm.c, ro, rd, gd, gm.tsdf, gm.tsdf_bnds, mini.bound, 'color', mini.n_samples, mini.n_surface,
                                   prefilter=(bound_dev, keep_b))
        assert torch.equal(keep_a, keep_b), case
        if case == 'mixed':
            assert 0 < int(keep_a.sum()) < keep_a.numel() and int(keep_a[5]) == 0
        else:
            assert int(keep_a.sum()) == 0 and float(dmax) == float('-inf')
        k = keep_a.bool()
        for x, y, what in zip(a[:4], b[:4], ('depth', 'uncertainty', 'colour', 'weight')):
            assert torch.equal(x[k], y[k]), (case, what)                                      # the kept rays: bit for bit
            assert torch.equal(torch.isnan(x), torch.isnan(y)), (case, what)
