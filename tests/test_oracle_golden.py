"""CPU: the oracle (oracle/adfp_oracle.py) against the golden vectors generated from the
reference's own modules (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import adfp_oracle as O


def _eq(a, b, tol=0.0):
    a = torch.as_tensor(a)
    b = torch.as_tensor(b)
    assert a.dtype == b.dtype, (a.dtype, b.dtype)
    assert a.shape == b.shape
    d = (a.double() - b.double()).abs().max().item() if a.numel() else 0.0
    assert d <= tol, d


@pytest.mark.parametrize('stage', O.STAGES)
def test_render_batch_ray_matches_reference(mini, stage):
    g = mini.golden(stage)
    d, u, c, w, aux = O.render_batch_ray(mini.sd, mini.c, mini.rays_d, mini.rays_o, mini.tsdf_volume, mini.tsdf_bnds,
                                         mini.bound, stage, mini.gt_depth, mini.n_samples, mini.n_surface,
                                         return_aux=True)
    _eq(aux['z_vals'], g['z_vals'])
    _eq(d, g['depth'], 1e-12)
    _eq(u, g['uncertainty'], 1e-12)
    _eq(c, g['color'], 1e-6)
    _eq(w, g['weight'], 1e-6)
    assert d.dtype == torch.float64 and u.dtype == torch.float64 and c.dtype == torch.float32
    assert tuple(w.shape) == (mini.rays_o.shape[0], mini.n_samples + mini.n_surface, 1)


@pytest.mark.parametrize('stage', O.STAGES)
def test_render_without_sensor_depth(mini, stage):
    g = mini.golden(stage)
    d, u, c, w, aux = O.render_batch_ray(mini.sd, mini.c, mini.rays_d, mini.rays_o, mini.tsdf_volume, mini.tsdf_bnds,
                                         mini.bound, stage, None, mini.n_samples, mini.n_surface, return_aux=True)
    _eq(aux['z_vals'], g['nd_z_vals'])
    _eq(d, g['nd_depth'], 1e-12)
    _eq(c, g['nd_color'], 1e-6)
    assert w.shape[1] == mini.n_samples


@pytest.mark.parametrize('stage', O.STAGES)
def test_eval_points_and_df_forward(mini, stage):
    g = mini.golden(stage)
    raw, w = O.eval_points(mini.sd, mini.query_points, mini.c, mini.tsdf_volume, mini.tsdf_bnds, mini.bound, stage)
    _eq(raw, g['q_raw'], 1e-6)
    _eq(w, g['q_w'], 1e-6)
    raw2, w2 = O.df_forward(mini.sd, mini.query_points, mini.c, mini.tsdf_volume, mini.tsdf_bnds, mini.bound, stage)
    _eq(raw2, g['df_raw'], 1e-6)
    _eq(w2, g['df_w'], 1e-6)
    # the query set really contains out-of-bound points (occ forced to 100, Renderer.py:64)
    assert (torch.as_tensor(g['q_raw'])[:, 3] == 100).any()


@pytest.mark.parametrize('stage', O.STAGES)
@pytest.mark.parametrize('tag,warm', [('g', False), ('gw', True)])
def test_mapper_loss_gradients(mini, stage, tag, warm):
    g = mini.golden(stage)
    c = {k: v.clone().requires_grad_(True) for k, v in mini.c.items()}
    sd = {k: v.clone().requires_grad_(True) for k, v in mini.sd.items()}
    d, u, col, w = O.render_batch_ray(sd, c, mini.rays_d, mini.rays_o, mini.tsdf_volume, mini.tsdf_bnds, mini.bound,
                                      stage, mini.gt_depth, mini.n_samples, mini.n_surface)
    loss = O.mapper_loss(d, col, w, mini.gt_depth, mini.gt_color, stage, warm)
    loss.backward()
    assert abs(loss.item() - float(g[tag + '.loss'])) <= 1e-9 * abs(float(g[tag + '.loss']))
    for k, v in c.items():
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        ref = torch.from_numpy(g[f'{tag}.{k}'])
        assert (got - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    for k, v in sd.items():
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        ref = torch.from_numpy(g[f'{tag}.sd.{k}'])
        assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()), k


def test_rays_tsdf_and_image(mini):
    g = mini.golden('rays')
    ro, rd = O.get_rays(mini.H, mini.W, mini.fx, mini.fy, mini.cx, mini.cy, mini.c2w)
    _eq(ro.contiguous(), g['get_rays_o'])
    _eq(rd, g['get_rays_d'])
    ro2, rd2 = O.get_rays_from_uv(torch.from_numpy(g['uv_i']), torch.from_numpy(g['uv_j']), mini.c2w,
                                  mini.fx, mini.fy, mini.cx, mini.cy)
    _eq(rd2, g['uv_rays_d'])
    t = O.trilerp(mini.tsdf_volume, mini.query_points, mini.tsdf_bnds)
    _eq(t, g['tsdf_q'])
    di, ui, ci = O.render_img(mini.sd, mini.c, mini.c2w, mini.H, mini.W, mini.fx, mini.fy, mini.cx, mini.cy,
                              mini.tsdf_volume, mini.tsdf_bnds, mini.bound, 'color', mini.depth_img,
                              mini.n_samples, mini.n_surface, ray_batch_size=int(g['img_ray_batch_size']))
    _eq(di, g['img_depth'], 1e-12)
    _eq(ui, g['img_uncertainty'], 1e-12)
    _eq(ci, g['img_color'], 1e-6)


def test_trilerp_explicit_is_grid_sample(mini):
    """The corner-by-corner restatement the HIP kernels implement == F.grid_sample, including
    points outside the volume (border clamp) and the permuted TSDF view."""
    a = O.trilerp(mini.tsdf_volume, mini.query_points, mini.tsdf_bnds)
    b = O.trilerp_explicit(mini.tsdf_volume, mini.query_points, mini.tsdf_bnds)
    assert (a - b).abs().max().item() <= 2e-7
    a = O.trilerp(mini.c['grid_high'], mini.query_points, mini.bound)
    b = O.trilerp_explicit(mini.c['grid_high'], mini.query_points, mini.bound)
    assert (a - b).abs().max().item() <= 1e-6 * a.abs().max().item()


def test_fixture_covers_edge_cases(mini):
    g = mini.golden('color')
    assert (mini.gt_depth == 0).sum() >= 5                      # zero-depth rays (Renderer.py:179-201)
    raw = torch.from_numpy(g['raw'])
    assert (raw[..., 3] == 100).any()                           # samples leaving the bound
    w = torch.from_numpy(g['weight'])
    frac = (w != 1).float().mean().item()
    assert 0.05 < frac < 0.95                                   # both in-band and out-of-band samples


def test_tracker_ray_gradients(mini):
    """d(Tracker loss)/d(rays) of the oracle == the reference's (tests/golden/mini_tracker.npz)."""
    g = mini.golden('tracker')
    ro = mini.rays_o.clone().requires_grad_(True)
    rd = mini.rays_d.clone().requires_grad_(True)
    d, u, c, w = O.render_batch_ray(mini.sd, mini.c, rd, ro, mini.tsdf_volume, mini.tsdf_bnds, mini.bound, 'color',
                                    mini.gt_depth, mini.n_samples, mini.n_surface)
    loss = O.tracker_loss(d, u, c, mini.gt_depth, mini.gt_color)
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= 1e-9 * abs(float(g['loss']))
    assert (ro.grad - torch.from_numpy(g['g_rays_o'])).abs().max().item() <= 1e-4
    assert (rd.grad - torch.from_numpy(g['g_rays_d'])).abs().max().item() <= 1e-4


def test_prefilter_mask_properties(mini):
    """a3 restatement (src/Mapper.py:438-449; the Mapper cannot be imported here, so this function is
    pinned by properties only): rays whose depth is the analytic ray/box distance are kept, rays
    reported 5 % beyond the bounding box are dropped."""
    from attentive_dfprior_amd import synthetic
    scene = synthetic.mini_scene()
    ro, rd, _, _ = synthetic.make_ray_batch(scene, 300, seed=4, zero_frac=0.0)
    t = (scene.bound.unsqueeze(0) - ro.unsqueeze(-1)) / rd.unsqueeze(-1)
    t_exit = torch.min(torch.max(t, dim=2)[0], dim=1)[0]
    assert bool(O.prefilter_mask(ro, rd, (0.95 * t_exit).float(), scene.bound).all())
    assert not bool(O.prefilter_mask(ro, rd, (1.05 * t_exit).float(), scene.bound).any())
    assert bool(O.prefilter_mask(ro, rd, torch.zeros(300), scene.bound).all())


# --------------------------------------------------------------------------- Mapper-side rows (a3, f4)
def _mapper_golden(name):
    import os
    import numpy as np
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def test_prefilter_mask_vs_reference_lines():
    """a3: tests/golden/mapper_prefilter.npz holds what the reference's OWN lines src/Mapper.py:438-449 computed
    (tests/golden/make_mapper_golden.py executes them); the oracle restatement must give the same mask, including
    the +-inf, NaN (0/0), non-finite-ray and depth == t cases."""
    g = _mapper_golden('mapper_prefilter.npz')
    names = sorted({k.split('.')[0] for k in g if '.' in k})
    assert 'exact' in names and len(names) >= 6
    for n in names:
        ro, rd, gd = (torch.from_numpy(g[f'{n}.{k}']) for k in ('rays_o', 'rays_d', 'gt_depth'))
        mask = O.prefilter_mask(ro, rd, gd, torch.from_numpy(g[f'{n}.bound']))
        assert torch.equal(mask, torch.from_numpy(g[f'{n}.inside_mask'])), n
    ex = torch.from_numpy(g['exact.inside_mask'])
    assert not ex[1] and not ex[2] and not ex[3]          # 0/0 on a bound plane: NaN compares false, ray dropped


def test_frustum_mask_vs_reference_lines():
    """f4: Mapper.get_mask_from_c2w (src/Mapper.py:90-158) executed from the reference's source with only
    cv2.remap substituted (OpenCV absent): the oracle's restatement of everything around the remap must reproduce
    those masks exactly (same remap on both sides)."""
    g = _mapper_golden('mapper_frustum.npz')
    H, W, fx, fy, cx, cy = g['intrinsics'].tolist()
    bound = torch.from_numpy(g['bound'])
    for k in range(3):
        c2w = torch.from_numpy(g[f'pose{k}.c2w'])
        depth = g[f'pose{k}.depth']
        for key in ('grid_low', 'grid_high', 'grid_color'):
            ref_xyz = g[f'pose{k}.{key}']                                     # [X, Y, Z] as the reference returns it
            X, Y, Z = ref_xyz.shape
            got = O.frustum_mask_np(c2w, (Z, Y, X), depth, bound, int(H), int(W), fx, fy, cx, cy)   # [Z, Y, X]
            assert (got == ref_xyz.transpose(2, 1, 0)).all(), (k, key)
            assert 0 < ref_xyz.sum() < ref_xyz.size


def test_frustum_boundary_set_names_the_points_on_a_decision_boundary():
    """oracle.frustum_boundary_points_np (the explicit list of grid points two correct float32 implementations may decide
    differently, tests/test_gpu_mapping.py): empty on a generic pose, and it does name a point that sits exactly on a boundary --
    a grid point at distance 0.5 from the camera centre (the ball of src/Mapper.py:146-151) and one at camera depth 0."""
    bound = torch.tensor([[0.0, 1.0], [0.0, 1.0], [0.0, 1.0]], dtype=torch.float64)
    depth = np.full((8, 8), 0.25, np.float32)
    c2w = torch.eye(4)                                                      # camera at the origin looking down -z
    boundary, mask = O.frustum_boundary_points_np(c2w, (3, 3, 3), depth, bound, 8, 8, 4.0, 4.0, 3.5, 3.5)
    assert (mask == O.frustum_mask_np(c2w, (3, 3, 3), depth, bound, 8, 8, 4.0, 4.0, 3.5, 3.5)).all()
    # grid points (x, y, z) in {0, 0.5, 1}^3, mask indexed [z, y, x]: (0.5, 0, 0), (0, 0.5, 0), (0, 0, 0.5) have dist^2 == 0.25
    for z, y, x in ((0, 0, 1), (0, 1, 0), (1, 0, 0)):
        assert boundary[z, y, x] and not mask[z, y, x]
    assert mask[0, 0, 0] and not boundary[0, 0, 0]                          # the camera centre itself: inside the ball, no doubt
    assert not boundary[2, 2, 2]
    g = _mapper_golden('mapper_frustum.npz')                                # the committed poses are generic: nothing may flip there
    H, W, fx, fy, cx, cy = g['intrinsics'].tolist()
    b, m = O.frustum_boundary_points_np(torch.from_numpy(g['pose0.c2w']), g['pose0.grid_high'].shape[::-1], g['pose0.depth'],
                                        torch.from_numpy(g['bound']), int(H), int(W), fx, fy, cx, cy)
    assert not b.any() and (m == g['pose0.grid_high'].transpose(2, 1, 0)).all()


def test_oracle_subnetworks_vs_reference_golden(mini):
    """mlp_forward / mlp_tsdf_forward against the REFERENCE's sub-modules called on their own
    (tests/golden/make_subnet_golden.py -> mini_subnets.npz; decoder.py:177-203, :240-258)."""
    g = mini.golden('subnets')
    qp = mini.query_points
    for name in ('low', 'high', 'color'):
        got = O.mlp_forward(mini.sd, name, qp, mini.c, mini.bound)
        assert got.shape == g[name].shape
        assert np.abs(got.numpy() - g[name]).max() == 0.0, name
        got32 = O.mlp_forward(mini.sd, name, qp.float(), mini.c, mini.bound)
        assert np.abs(got32.numpy() - g[name + '_f32']).max() == 0.0, name
    t = O.trilerp(mini.tsdf_volume, qp, mini.tsdf_bnds).reshape(-1)
    fused, w = O.mlp_tsdf_forward(mini.sd, torch.from_numpy(g['att_occ_in']), t)
    assert np.abs(fused.numpy() - g['att_fused']).max() <= 1e-6
    assert np.abs(w.numpy() - g['att_w']).max() <= 1e-6


def test_pose_utilities_match_the_reference():
    """common.quad2rotation / get_camera_from_tensor (the host-side pose utilities of the Tracker) and their autograd against what the
    reference's own functions returned (tests/golden/mini_pose.npz, src/common.py:139-178); get_tensor_from_camera inverts them (the
    reference's goes through mathutils, which the image lacks: pinned by the round trip instead)."""
    import os
    from attentive_dfprior_amd import common
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'mini_pose.npz'))
    cam, cot = torch.from_numpy(g['cam']), torch.from_numpy(g['cot'])
    assert np.abs(common.get_camera_from_tensor(cam).numpy() - g['c2w_batched']).max() <= 1e-6
    for k in range(cam.shape[0]):
        t = cam[k].clone().requires_grad_(True)
        RT = common.get_camera_from_tensor(t)
        (RT * cot[k]).sum().backward()
        assert np.abs(RT.detach().numpy() - g['c2w'][k]).max() <= 1e-6 * max(1.0, np.abs(g['c2w'][k]).max())
        assert np.abs(t.grad.numpy() - g['g_cam'][k]).max() <= 2e-5 * max(1.0, np.abs(g['g_cam'][k]).max())
        back = common.get_camera_from_tensor(common.get_tensor_from_camera(torch.from_numpy(g['c2w'][k])))
        assert np.abs(back.numpy() - g['c2w'][k]).max() <= 1e-5 * max(1.0, np.abs(g['c2w'][k]).max())


def _oracle_grads(mini, stage, warm, relu_masks=None):
    c_or = {k: v.clone().requires_grad_(True) for k, v in mini.c.items()}
    sd_or = {k: v.clone().requires_grad_(True) for k, v in mini.sd.items()}
    d2, u2, col2, w2 = O.render_batch_ray(sd_or, c_or, mini.rays_d, mini.rays_o, mini.tsdf_volume, mini.tsdf_bnds, mini.bound, stage,
                                          mini.gt_depth, mini.n_samples, mini.n_surface, relu_masks=relu_masks)
    loss2 = O.mapper_loss(d2, col2, w2, mini.gt_depth, mini.gt_color, stage, warm)
    loss2.backward()
    zero = lambda t: t.grad if t.grad is not None else torch.zeros_like(t)           # noqa: E731
    return loss2, {k: zero(v) for k, v in c_or.items()}, {k: zero(v) for k, v in sd_or.items()}


def test_forced_relu_decisions_equal_to_relus_own_change_nothing(mini):
    """The test-only oracle variant is the oracle: forcing the decisions relu itself takes reproduces its gradients bit for bit
    (so the tight comparisons above are comparisons with the pinned oracle, moved only where a kernel decided a boundary unit
    the other way)."""
    loss, gc, gsd = _oracle_grads(mini, 'color', True)
    # relu's own decisions, recorded from a plain forward
    P = mini.rays_o.shape[0] * (mini.n_samples + mini.n_surface)
    rec = {}
    orig = O.F.relu

    def recording(h):
        rec.setdefault('calls', []).append(h.detach() > 0)
        return orig(h)
    O.F.relu = recording
    try:
        with torch.no_grad():
            O.render_batch_ray(mini.sd, mini.c, mini.rays_d, mini.rays_o, mini.tsdf_volume, mini.tsdf_bnds, mini.bound, 'color',
                               mini.gt_depth, mini.n_samples, mini.n_surface)
    finally:
        O.F.relu = orig
    calls = rec['calls']                                  # low x5, high x5, color x5 (P rows each), att x4 (band rows)
    assert len(calls) == 19 and all(m.shape[0] == P for m in calls[:15])
    with torch.no_grad():
        _, _, _, _, aux = O.render_batch_ray(mini.sd, mini.c, mini.rays_d, mini.rays_o, mini.tsdf_volume, mini.tsdf_bnds, mini.bound,
                                             'color', mini.gt_depth, mini.n_samples, mini.n_surface, return_aux=True)
    band = aux['band']
    att = []
    for m in calls[15:]:
        full = torch.zeros((P, m.shape[1]), dtype=torch.bool)
        full[band] = m
        att.append(full)
    rm = {'low': torch.stack(calls[0:5], 1), 'high': torch.stack(calls[5:10], 1), 'high_valid': torch.ones(P, dtype=torch.bool),
          'color': torch.stack(calls[10:15], 1), 'att': att, 'band': band}
    O.reset_relu_flips()
    loss2, gc2, gsd2 = _oracle_grads(mini, 'color', True, relu_masks=rm)
    assert O.RELU_FLIPS['flipped'] == 0 and O.RELU_FLIPS['units'] > 0
    assert torch.equal(loss, loss2)
    for k in gc:
        assert torch.equal(gc[k], gc2[k]), k
    for k in gsd:
        assert torch.equal(gsd[k], gsd2[k]), k
