"""GPU: the RCCL path of attentive_dfprior_amd.dist on the device (backend "nccl" IS RCCL on ROCm), in a world of ONE
rank -- every GPU test box has one MI355X; the N > 1 logic is pinned by tests/test_dist_gloo.py (gloo, world_size 2)
and the N-GPU run is `python bench.py --gpus N`.  What this covers on real hardware: the process group comes up on
the HIP device, device tensors go through the flat-bucket all-reduce / the frustum-masked bucket / the output
all-gather of render_rays_sharded, and the results are what a single rank must get (unchanged gradients, the
unsharded render)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

import attentive_dfprior_amd as A
from attentive_dfprior_amd import dist as adist
from conftest import make_cfg, to_dev, assert_adam_trajectory

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300, method='thread')]
DEV = torch.device('cuda:0')


@pytest.fixture(scope='module')
def rccl():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(DEV)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=DEV)
    yield dist
    dist.destroy_process_group()


def test_rccl_world_is_up(rccl):
    assert rccl.get_backend() == 'nccl' and rccl.get_world_size() == 1
    t = torch.full((1024,), 3.0, device=DEV)
    rccl.all_reduce(t)
    torch.cuda.synchronize()
    assert float(t.sum()) == 3.0 * 1024


def test_gradient_buckets_through_rccl(rccl, mini):
    dec = A.DF()
    dec.load_state_dict(mini.sd)
    dec.bound = mini.bound
    dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(mini.n_samples, mini.n_surface), None, mini)
    c = {k: v.to(DEV).requires_grad_(True) for k, v in mini.c.items()}
    ro, rd, gd, gc = mini.rays_o.to(DEV), mini.rays_d.to(DEV), mini.gt_depth.to(DEV), mini.gt_color.to(DEV)
    d, u, col, w = rend.render_batch_ray(c, dec, rd, ro, DEV, mini.tsdf_volume.to(DEV), mini.tsdf_bnds.to(DEV), 'color', gd)
    m = gd > 0
    (torch.abs(gd[m] - d[m]).sum() + 0.2 * torch.abs(gc - col).sum()).backward()
    tensors = list(c.values()) + list(dec.parameters())
    before = [t.grad.clone() for t in tensors]
    nbytes = adist.allreduce_grads(tensors, skip_single=False)          # world of one: SUM leaves every gradient as it was
    torch.cuda.synchronize()
    assert nbytes == 4 * sum(t.numel() for t in tensors)
    for a, b in zip(before, tensors):
        assert torch.equal(a, b.grad)
    assert adist.allreduce_grads(tensors) == 0                         # default: skipped in a world of one
    masks = {k: (torch.rand(v.shape[2:], device=DEV) < 0.3) for k, v in c.items()}
    bucket = adist.MaskedGradBucket(c, masks, extra=list(dec.parameters()))
    assert bucket.numel() < sum(t.numel() for t in tensors)
    bucket.allreduce(skip_single=False)
    torch.cuda.synchronize()
    for a, b in zip(before, tensors):
        assert torch.equal(a, b.grad)


def test_sharded_render_gathers_through_rccl(rccl, mini):
    dec = A.DF()
    dec.load_state_dict(mini.sd)
    dec.bound = mini.bound
    dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(mini.n_samples, mini.n_surface), None, mini)
    c = to_dev(mini.c, DEV)
    tsdf, tb = mini.tsdf_volume.to(DEV), mini.tsdf_bnds.to(DEV)
    ro, rd, gd = mini.rays_o.to(DEV), mini.rays_d.to(DEV), mini.gt_depth.to(DEV)

    def render_fn(o, d_, z, mx):
        return rend.render_batch_ray(c, dec, d_, o, DEV, tsdf, tb, 'color', z, depth_max=mx)
    with torch.no_grad():
        outs = adist.render_rays_sharded(render_fn, ro, rd, gd)
        whole = rend.render_batch_ray(c, dec, rd, ro, DEV, tsdf, tb, 'color', gd)
        # the gather itself, on device tensors, as the N > 1 path issues it
        rows = adist._all_gather_rows(whole[2].contiguous(), [whole[2].shape[0]], None)
    for a, b in zip(outs, whole):
        assert torch.equal(a, b)
    assert torch.equal(rows, whole[2])
    gmax = adist.global_depth_max(gd)
    assert float(gmax) == float(gd.max())


@pytest.mark.parametrize('masked', [False, True])
def test_fused_mapper_iteration_reduces_its_bucket_through_rccl(rccl, mini, masked):
    """mapping.MapperIteration in distributed mode: the backward writes into slices of one contiguous bucket, the stage's
    prefix of it goes through ONE all-reduce (plus a one-float MAX for the far clamp).  In a world of one rank the result
    must equal the non-distributed iteration.  With frustum masks only the selected voxels' gradient columns travel."""
    import copy
    from attentive_dfprior_amd import mapping, synthetic
    sc = synthetic.mini_scene()
    lr = {s: dict(low=0.01, high=0.005, color=0.005, decoders=0.005, mlp=0.005) for s in ('low', 'high', 'color')}
    tsdf, tb = sc.tsdf_volume.to(DEV), sc.tsdf_bnds.to(DEV)
    rays = [t.to(DEV) for t in synthetic.make_ray_batch(sc, 500, seed=3)]
    masks = None
    if masked:
        c2w = sc.default_c2w(yaw=0.7, pitch=0.1)
        masks = {k: mapping.frustum_mask(c2w, tuple(v.shape[2:]), sc.depth_image(c2w).to(DEV), sc.bound, sc.H, sc.W, sc.fx, sc.fy,
                                         sc.cx, sc.cy) for k, v in sc.c.items()}
    res = []
    for distributed in (True, False):
        dec = A.DF()
        dec.load_state_dict(mini.sd)
        dec.bound = sc.bound
        dec = dec.to(DEV)
        grids = {k: v.clone().to(DEV) for k, v in sc.c.items()}
        it = mapping.MapperIteration(A.Renderer(make_cfg(32, 16), None, sc), dec, grids, masks, tsdf, tb, lr, use_graph=False,
                                     distributed=distributed)
        losses = [float(it.step(*rays, stage)) for stage in ('low', 'high', 'color', 'color')]
        if distributed and not masked:
            assert it.bucket_bytes == 4 * it.bucket.numel()                # stage color reduces the whole bucket
        if distributed and masked:
            sel = sum(int(m.sum()) * sc.c[k].shape[1] for k, m in masks.items())
            assert it.bucket_bytes == 4 * (sel + sum(f.numel() for f in it.flat.values())) < 4 * it.bucket.numel()
        res.append((grids, dec, losses))
    (ga, da, la), (gb, db, lb) = res
    assert all(abs(x - y) <= 1e-5 * abs(y) for x, y in zip(la, lb)), (la, lb)       # later losses inherit the Adam noise below
    for k in ga:                                   # float atomics: the two runs differ in the last bits, Adam amplifies noise-sized gradients
        assert_adam_trajectory(ga[k], gb[k], 0.01, 4, k, max_outliers=1e-2)
    for (n, p), (_, q) in zip(da.named_parameters(), db.named_parameters()):
        # ReLU-boundary samples, Adam-amplified (conftest).  The two runs differ by the ORDER of float atomics only, and a 128-element
        # bias has a handful of units whose gradient is noise-sized: 7 of 128 left the trajectory in one run of eight (5.5 %)
        assert_adam_trajectory(p, q, 0.005, 4, n, max_outliers=8e-2)
