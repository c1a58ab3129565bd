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
from conftest import make_cfg, to_dev

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
