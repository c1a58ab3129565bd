"""GPU: the Mapper bookkeeping kernels -- frustum feature mask against the numpy
restatement of Mapper.get_mask_from_c2w (pinned by the reference's own lines with cv2.remap substituted: cv2 is absent,
see oracle header) and the masked in-place Adam against torch.optim.Adam on the compact copy the reference optimises.

A grid point whose mask differs from the oracle's is accepted ONLY when it is in the oracle's explicit boundary set
(oracle.frustum_boundary_points_np: its decision changes when its camera-space coordinates move within the float32
summation-order bound of `w2c @ homo`, the one operation whose order src/Mapper.py:116-117 leaves to numpy) -- rounds 3-4 tolerated
a COUNT of flips ("<= 2") without saying which points may flip."""
import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import mapping, synthetic
from oracle import adfp_oracle as O
from conftest import make_cfg

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def assert_only_boundary_points_differ(got, ref, boundary, what):
    """Every grid point outside the oracle's boundary set must agree; the set itself must stay a sliver of the grid (a criterion
    that excuses 1 % of the points would excuse a wrong kernel)."""
    import numpy as np
    flips = got != ref
    assert boundary.sum() <= max(2, ref.size // 2000), f'{what}: {int(boundary.sum())} of {ref.size} grid points are boundary points'
    stray = flips & ~boundary
    assert not stray.any(), (f'{what}: {int(stray.sum())} grid points differ from the oracle away from every decision boundary, '
                             f'first at [z, y, x] = {np.argwhere(stray)[0].tolist()} ({int(flips.sum())} differ in all, '
                             f'{int(boundary.sum())} boundary points)')


@pytest.mark.parametrize('scene_name,pose', [('mini', dict()), ('mini', dict(offset=(0.3, -0.2, 0.1), yaw=2.1, pitch=0.4)),
                                             ('room0', dict(yaw=1.0, pitch=-0.2))])
def test_frustum_mask_vs_oracle(scene_name, pose):
    sc = synthetic.mini_scene() if scene_name == 'mini' else synthetic.Scene('room0', device='cpu')
    c2w = sc.default_c2w(**pose)
    depth = sc.depth_image(c2w)
    for key, val in sc.c.items():
        shp = tuple(val.shape[2:])
        ref = O.frustum_mask_np(c2w, shp, depth.numpy(), sc.bound, sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy)
        boundary, again = O.frustum_boundary_points_np(c2w, shp, depth.numpy(), sc.bound, sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy)
        assert (again == ref).all()
        got = mapping.frustum_mask(c2w, shp, depth.to(DEV), sc.bound, sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy)
        assert got.dtype == torch.bool and tuple(got.shape) == shp
        assert_only_boundary_points_differ(got.cpu().numpy(), ref, boundary, key)
        assert 0 < ref.sum() < ref.size


@pytest.mark.parametrize('masked', [True, False])
def test_masked_adam_vs_torch_adam(masked):
    g = torch.Generator().manual_seed(3)
    shape = (1, 32, 5, 7, 9)                       # nvox = 315: exercises the ragged tail of the 4-voxel threads
    p0 = torch.randn(shape, generator=g) * 0.01
    mask = (torch.rand(shape[2:], generator=g) < 0.4) if masked else None
    grads = [torch.randn(shape, generator=g) * (10.0 ** -k) for k in range(4)]
    lrs = [0.1, 0.005, 0.0, 0.005]
    # reference: Adam on the compact copy (src/Mapper.py:347-378)
    full = torch.ones(shape, dtype=torch.bool) if mask is None else mask[None, None].expand(shape)
    val_grad = p0[full].clone().requires_grad_(True)
    opt = torch.optim.Adam([{'params': [val_grad], 'lr': 0}])
    # product
    grid = p0.clone().to(DEV).requires_grad_(True)
    mine = mapping.MaskedGridAdam({'g': grid}, {'g': mask})
    for gr, lr in zip(grads, lrs):
        opt.param_groups[0]['lr'] = lr
        val_grad.grad = gr[full].clone()
        opt.step()
        v0 = grid._version
        grid.grad = gr.to(DEV)
        mine.step({'g': lr})
        assert grid._version > v0                  # layout caches keyed on _version must see the update
    out = grid.detach().cpu()
    ref = p0.clone()
    ref[full] = val_grad.detach()
    assert torch.equal(out[~full], p0[~full])      # untouched outside the mask, bit for bit
    err = ((out - ref).abs() / ref.abs().clamp_min(1e-3)).max()
    assert float(err) <= 2e-6, float(err)


def test_mapping_iterations_equal_reference_style_loop():
    """Three mapping iterations with frustum-masked grids.  The reference's loop keeps a compact val_grad as
    the Adam parameter and index_puts it into the grid before every render (src/Mapper.py:347-388); the
    in-place masked Adam must walk the same trajectory.  Adam turns the SIGN of a gradient into a step of
    size lr, so the two loops are fed the same gradient values (float atomics make the last bits of two
    backward passes differ, which flips the sign of gradients that cancel to ~0); that the two loops'
    own gradients agree to atomics noise is asserted separately."""
    sc = synthetic.mini_scene()
    sd = O.random_state_dict(seed=3)
    ro, rd, gd, gc = (t.to(DEV) for t in synthetic.make_ray_batch(sc, 200, seed=2))
    c2w = sc.default_c2w()
    depth_img = sc.depth_image(c2w).to(DEV)
    masks = {k: mapping.frustum_mask(c2w, tuple(v.shape[2:]), depth_img, sc.bound, sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy)
             for k, v in sc.c.items()}
    rend = A.Renderer(make_cfg(32, 16), None, sc)
    tsdf, bnds = sc.tsdf_volume.to(DEV), sc.tsdf_bnds.to(DEV)
    lrs = {'grid_low': 0.1, 'grid_high': 0.005, 'grid_color': 0.005}
    dec = A.DF()
    dec.load_state_dict(sd)
    dec.bound = sc.bound
    for p in dec.parameters():
        p.requires_grad_(False)
    dec = dec.to(DEV)

    def loss_of(c):
        d, u, col, w = rend.render_batch_ray(c, dec, rd, ro, DEV, tsdf, bnds, 'color', gt_depth=gd)
        m = gd > 0
        return torch.abs(gd[m] - d[m]).sum() + 0.2 * torch.abs(gc - col).sum()

    c_new = {k: v.clone().to(DEV).requires_grad_(True) for k, v in sc.c.items()}
    mine = mapping.MaskedGridAdam(c_new, masks)
    c_ref = {k: v.clone().to(DEV) for k, v in sc.c.items()}
    full = {k: masks[k][None, None].expand(c_ref[k].shape) for k in c_ref}
    vg = {k: c_ref[k][full[k]].clone().requires_grad_(True) for k in c_ref}
    opt = torch.optim.Adam([{'params': [vg[k]], 'lr': lrs[k]} for k in c_ref])
    for _ in range(3):
        # in place: leaf grids, dense gradient, masked Adam
        mine.zero_grad()
        loss_of(c_new).backward()
        grads = {k: c_new[k].grad.clone() for k in c_new}
        mine.step(lrs)
        # reference style: index_put the compact parameter into the grid, render, backward through the index_put
        c_it = {}
        for k in c_ref:
            val = c_ref[k].clone()
            val[full[k]] = vg[k]
            c_it[k] = val
        opt.zero_grad()
        loss_of(c_it).backward()
        for k in c_ref:
            own = vg[k].grad
            assert float((own - grads[k][full[k]]).abs().max()) <= 1e-4 * float(grads[k].abs().max()) + 1e-12, k
            vg[k].grad = grads[k][full[k]].clone()
        opt.step()
    for k in c_ref:
        c_ref[k][full[k]] = vg[k].detach()
        a, b = c_new[k].detach(), c_ref[k]
        assert bool(full[k].any()) and not bool(full[k].all())
        assert torch.equal(a[~full[k]], b[~full[k]])
        assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max()), k      # a few ulp of the lr-sized steps


def test_frustum_mask_vs_reference_lines():
    """HIP frustum mask against the masks the reference's own get_mask_from_c2w lines produced
    (tests/golden/mapper_frustum.npz; only cv2.remap substituted)."""
    import os
    import numpy as np
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, 'mapper_frustum.npz'))
    H, W, fx, fy, cx, cy = g['intrinsics'].tolist()
    bound = torch.from_numpy(g['bound'])
    for k in range(3):
        c2w = torch.from_numpy(g[f'pose{k}.c2w'])
        depth = torch.from_numpy(g[f'pose{k}.depth'])
        for key in ('grid_low', 'grid_high', 'grid_color'):
            ref = g[f'pose{k}.{key}'].transpose(2, 1, 0)                      # [X,Y,Z] -> the grid tensor's [Z,Y,X]
            boundary, again = O.frustum_boundary_points_np(c2w, ref.shape, depth.numpy(), bound, int(H), int(W), fx, fy, cx, cy)
            assert (again == ref).all()                                       # the oracle IS the reference-executed mask
            got = mapping.frustum_mask(c2w, ref.shape, depth.to(DEV), bound, int(H), int(W), fx, fy, cx, cy)
            assert_only_boundary_points_differ(got.cpu().numpy(), ref, boundary, f'pose {k} {key}')


def test_adam_step_equals_the_two_entries_it_merges():
    """adfp_adam_step (channels-last grids and flat network buffers in ONE launch) against adfp_adam_grids_cl + adfp_masked_adam_multi:
    every tensor bit for bit over three steps -- parameters (shadow and reference-layout master), both moments, the consumed
    gradients zeroed; with masks, a group without a mask, and a group that does not step (derived = 0, the skipped iteration)."""
    import ctypes as C
    from attentive_dfprior_amd import _lib
    from attentive_dfprior_amd._lib import lib, ptr, check
    L = lib()
    st = _lib.current_stream(torch.device(DEV))
    g = torch.Generator().manual_seed(8)
    shapes = [(5, 6, 7), (9, 4, 11), (3, 3, 3)]
    flats = [1000, 37, 4097]

    def make():
        G = torch.Generator().manual_seed(9)
        cl = []
        for k, (Z, Y, X) in enumerate(shapes):
            nv = Z * Y * X
            cl.append(dict(p_cl=torch.randn(nv, 32, generator=G).to(DEV), p_cm=torch.zeros(32, nv, device=DEV), g=None,
                           m=torch.zeros(nv, 32, device=DEV), v=torch.zeros(nv, 32, device=DEV),
                           mask=None if k == 1 else (torch.rand(nv, generator=G) < 0.4).to(DEV, torch.uint8), nv=nv))
            cl[-1]['p_cm'].copy_(cl[-1]['p_cl'].t())
        fl = [dict(p=torch.randn(n, generator=G).to(DEV), m=torch.zeros(n, device=DEV), v=torch.zeros(n, device=DEV), n=n) for n in flats]
        derived = torch.zeros(len(shapes) + len(flats), 2, device=DEV)
        return cl, fl, derived
    A_, B_ = make(), make()
    for it in range(3):
        grads_cl = [torch.randn(s[0] * s[1] * s[2], 32, generator=g) for s in shapes]
        grads_fl = [torch.randn(n, generator=g) for n in flats]
        t = it + 1
        der = torch.tensor([[1e-2 / (1 - 0.9 ** t), (1 - 0.999 ** t) ** 0.5]] * (len(shapes) + len(flats)))
        der[2] = 0.0                                                       # the third grid does not step
        for merged, (cl, fl, derived) in ((False, A_), (True, B_)):
            derived.copy_(der.to(DEV))
            carr = (_lib.AdfpAdamClGroup * len(cl))()
            gcl = [x.clone().to(DEV) for x in grads_cl]
            gfl = [x.clone().to(DEV) for x in grads_fl]
            for k, c in enumerate(cl):
                a = carr[k]
                a.param_cl, a.param_cm, a.grad_cl = c['p_cl'].data_ptr(), c['p_cm'].data_ptr(), gcl[k].data_ptr()
                a.exp_avg_cl, a.exp_avg_sq_cl = c['m'].data_ptr(), c['v'].data_ptr()
                a.mask = c['mask'].data_ptr() if c['mask'] is not None else None
                a.nvox, a.derived = c['nv'], derived[k].data_ptr()
            farr = (_lib.AdfpAdamGroup * len(fl))()
            for k, f in enumerate(fl):
                a = farr[k]
                a.param, a.grad, a.exp_avg, a.exp_avg_sq = f['p'].data_ptr(), gfl[k].data_ptr(), f['m'].data_ptr(), f['v'].data_ptr()
                a.mask, a.nvox, a.channels, a.derived = None, f['n'], 1, derived[len(cl) + k].data_ptr()
            if merged:
                check(L.adfp_adam_step(len(cl), C.byref(carr), len(fl), C.byref(farr), 0.9, 0.999, 1e-8, st), 'adfp_adam_step')
            else:
                check(L.adfp_adam_grids_cl(len(cl), C.byref(carr), 0.9, 0.999, 1e-8, st), 'adfp_adam_grids_cl')
                check(L.adfp_masked_adam_multi(len(fl), C.byref(farr), 0.9, 0.999, 1e-8, st), 'adfp_masked_adam_multi')
            torch.cuda.synchronize()
            for x in gcl:
                assert not x.any()                                         # consumed and zeroed
        for ca, cb in zip(A_[0], B_[0]):
            for key in ('p_cl', 'p_cm', 'm', 'v'):
                assert torch.equal(ca[key], cb[key]), (it, key)
        for fa, fb in zip(A_[1], B_[1]):
            for key in ('p', 'm', 'v'):
                assert torch.equal(fa[key], fb[key]), (it, key)
    assert A_[0][0]['m'].any() and not A_[0][2]['m'].any()                 # the stepping grid moved, the skipped one did not

