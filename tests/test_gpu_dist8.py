"""GPU: BASELINE.json configs[3] at WORLD SIZE 8 on one MI355X -- eight processes share the device and talk through gloo (every
test box has ONE GPU, so RCCL itself only ever runs at world size 1: tests/test_gpu_nccl.py; the collectives here are the same
torch.distributed calls on device tensors, and everything around them -- ragged contiguous shards, the full-batch far clamp, the
pack kernel / ONE all-gather / unpack kernel of the frame's outputs, the in-place bucket all-reduce of the gradients that
autograd holds as views of one buffer, the frustum-masked compact bucket -- is the code the 8-GPU run executes).

  scene0050 bounds, 620 x 460 = 285 200 rays (35 650 per rank) and a RAGGED batch of 100 003 rays (12 500 / 12 501 per rank):
      the gathered render = the single-process render, bit for bit
  5 000 rays of the Mapper (scannet.yaml mapping.pixels), colour stage, colour decoder + attention MLP trainable, dense grids:
      sum over ranks (dist.allreduce_grads, zero copy) = the single-process gradient to float summation order; the same through
      dist.MaskedGradBucket with frustum masks (only the selected voxels travel)
"""
import os
import socket
import subprocess
import sys
import tempfile

import pytest
import torch

from conftest import assert_close, assert_close_scale      # sums over shards in a different order: held to each tensor's scale, AND element by element

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
WORLD = 8

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import torch
import torch.distributed as dist
import attentive_dfprior_amd as A
from attentive_dfprior_amd import synthetic, mapping, dist as adist
from attentive_dfprior_amd.common import get_rays
from conftest import make_cfg

rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
DEV = torch.device('cuda:0')
if world > 1:
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:' + port, rank=rank, world_size=world)
sc = synthetic.Scene('scene0050', H=460, W=620, fx=577.590698, fy=578.729797, cx=318.905426 - 10, cy=242.683609 - 10,
                     device=DEV, grid_std_scale=20.0, inset=0.4)
sc.c['grid_high'] = sc.c['grid_high'] * 100
dec = A.DF(); dec.load_state_dict(synthetic.seeded_state_dict(0)); dec.bound = sc.bound; dec = dec.to(DEV)
rend = A.Renderer(make_cfg(48, 16), None, sc, ray_batch_size=10 ** 9)
tb = sc.tsdf_bnds.to(DEV)
c2w = sc.default_c2w(yaw=0.4, pitch=-0.15)
gd = sc.depth_image(c2w).reshape(-1)
ro, rd = get_rays(sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, c2w, DEV)
ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
res = {}

def render_fn(o, d, z, m):
    return rend.render_batch_ray(sc.c, dec, d, o, DEV, sc.tsdf_volume, tb, 'color', z, depth_max=m)[:3]

with torch.no_grad():
    for name, n in (('frame', ro.shape[0]), ('ragged', 100003)):
        outs = adist.render_rays_sharded(render_fn, ro[:n], rd[:n], gd[:n], gather=True)
        if rank == 0:
            res[name] = [t.cpu() for t in outs]

# ---- Renderer.render_img, ray-sharded: the reference's 100 000-ray batches (their far clamps) cut across the 35 650-ray shards
rend_b = A.Renderer(make_cfg(48, 16), None, sc)
assert rend_b.ray_batch_size == 100000
with torch.no_grad():
    imgs = adist.render_img_sharded(rend_b, sc.c, dec, c2w, DEV, sc.tsdf_volume, tb, 'color', sc.depth_image(c2w))
    if rank == 0:
        res['img'] = [t.cpu() for t in imgs]
        if world == 1:
            res['img_ref'] = [t.cpu() for t in rend_b.render_img(sc.c, dec, c2w, DEV, sc.tsdf_volume, tb, 'color', gt_depth=sc.depth_image(c2w))]

# ---- the Mapper's 5 000 rays: every rank differentiates ITS shard, the gradients are summed over the ranks
for p in list(dec.low_decoder.parameters()) + list(dec.high_decoder.parameters()):
    p.requires_grad_(False)
params = [p for p in dec.parameters() if p.requires_grad]
g = torch.Generator().manual_seed(3)
pick = torch.randint(ro.shape[0], (5000,), generator=g).to(DEV)
mo, md, mz = ro[pick].contiguous(), rd[pick].contiguous(), gd[pick].contiguous()
mc = torch.rand(5000, 3, generator=g).to(DEV)
dmax = mz.max().reshape(1)
lo, hi = adist.shard_range(5000, rank, world)
masks = {k: mapping.frustum_mask(c2w, tuple(v.shape[2:]), sc.depth_image(c2w), sc.bound, sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy) for k, v in sc.c.items()}
for mode in ('dense', 'masked'):
    c = {k: v.detach().clone().requires_grad_(True) for k, v in sc.c.items()}
    for p in params:
        p.grad = None
    d, u, col, w = rend.render_batch_ray(c, dec, md[lo:hi], mo[lo:hi], DEV, sc.tsdf_volume, tb, 'color', mz[lo:hi], depth_max=dmax)
    m = mz[lo:hi] > 0
    (torch.abs(mz[lo:hi][m] - d[m]).sum() + 0.2 * torch.abs(mc[lo:hi] - col).sum()).backward()
    tensors = list(c.values()) + params
    if mode == 'dense':
        zero_copy = adist._common_bucket([t.grad for t in tensors]) is not None
        nbytes = adist.allreduce_grads(tensors) if world > 1 else sum(t.numel() for t in tensors) * 4
    else:
        zero_copy = None
        bucket = adist.MaskedGradBucket(c, masks, extra=params)
        bucket.allreduce()
        nbytes = bucket.numel() * 4
    if rank == 0:
        grads = {k: v.grad.cpu() for k, v in c.items()}
        if mode == 'masked':                                   # outside the mask a rank keeps its LOCAL gradient (never used): compare inside
            grads = {k: torch.where(masks[k].cpu().reshape((1, 1) + tuple(masks[k].shape)), v, torch.zeros_like(v)) for k, v in grads.items()}
        res[mode] = {'grids': grads, 'params': [p.grad.cpu() for p in params], 'bytes': nbytes, 'zero_copy': zero_copy,
                     'masks': {k: v.cpu() for k, v in masks.items()}}
if rank == 0:
    torch.save(res, out)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return str(s.getsockname()[1])


def _run(world, tmp):
    out = os.path.join(tmp, f'w{world}.pt')
    port = _free_port()
    code = WORKER % {'root': ROOT, 'here': HERE}
    procs = [subprocess.Popen([sys.executable, '-c', code, str(r), str(world), port, out], env=dict(os.environ)) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    return torch.load(out)


@pytest.fixture(scope='module')
def runs():
    with tempfile.TemporaryDirectory() as tmp:
        return _run(1, tmp), _run(WORLD, tmp)


def test_eight_ranks_gather_the_single_process_render(runs):
    one, eight = runs
    for name in ('frame', 'ragged'):
        assert eight[name][0].shape[0] == {'frame': 620 * 460, 'ragged': 100003}[name]
        for k, what in enumerate(('depth', 'uncertainty', 'colour')):
            assert eight[name][k].dtype == one[name][k].dtype
            assert torch.equal(eight[name][k], one[name][k]), f'{name}: gathered {what} differs from the single-process render'


def test_eight_ranks_render_img_sharded_equals_render_img(runs):
    """dist.render_img_sharded (the bench's N > 1 headline): eight contiguous pixel ranges, the far clamp of every 100 000-ray batch
    taken over the whole frame, ONE packed all-gather -- the images Renderer.render_img returns, bit for bit."""
    one, eight = runs
    for k, what in enumerate(('depth', 'uncertainty', 'colour')):
        ref = one['img_ref'][k]
        assert ref.shape[:2] == (460, 620)
        assert one['img'][k].dtype == ref.dtype and eight['img'][k].dtype == ref.dtype
        assert torch.equal(one['img'][k], ref), f'{what}: one rank, shard = the whole frame'
        assert torch.equal(eight['img'][k], ref), f'{what}: eight ranks gathered'


def test_eight_ranks_sum_to_the_single_process_gradient(runs):
    one, eight = runs
    assert eight['dense']['zero_copy'] is True              # autograd's .grad tensors are views of ONE buffer: all-reduced in place
    assert 6.0e6 < eight['dense']['bytes'] < 8.0e6           # scene0050 ~ 6.8 MB dense
    assert eight['masked']['bytes'] < eight['dense']['bytes']
    for mode in ('dense', 'masked'):
        for k in one[mode]['grids']:
            ref = one[mode]['grids'][k]
            if mode == 'masked':
                m = one['masked']['masks'][k]
                ref = torch.where(m.reshape((1, 1) + tuple(m.shape)), one['dense']['grids'][k], torch.zeros_like(ref))
            assert_close_scale(eight[mode]['grids'][k], ref, 2e-5, f'{mode}: sum of 8 shard gradients of {k}')
            # ... and per element (relative down to 1 % of the scale): an error confined to small-magnitude elements -- one voxel
            # column owned by the wrong rank, a masked column that did not travel -- passes a scale-relative bound
            assert_close(eight[mode]['grids'][k], ref, 1e-4, f'{mode}: sum of 8 shard gradients of {k}, per element')
        for a, b in zip(eight[mode]['params'], one['dense']['params']):
            assert_close_scale(a, b, 2e-5, f'{mode}: sum of 8 shard parameter gradients')
            assert_close(a, b, 1e-4, f'{mode}: sum of 8 shard parameter gradients, per element')
