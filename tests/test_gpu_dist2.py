"""GPU: the ray-sharded fused Mapper iteration at WORLD SIZE 2 on one MI355X -- two processes share the device and talk through
gloo (every test box has one GPU, so RCCL itself can only be exercised at world size 1: tests/test_gpu_nccl.py; the collectives
here are the same torch.distributed calls on device tensors).  Each rank steps mapping.MapperIteration(distributed=True) on its
half of the rays: the far clamp is the MAX over ranks, the gradient bucket -- compact when frustum-masked -- the SUM.  The
Mapper's losses are plain sums over rays, so the result must follow the single-process iteration over all rays."""
import os
import socket
import subprocess
import sys
import tempfile

import pytest
import torch

from conftest import assert_adam_trajectory

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import torch
import torch.distributed as dist
import attentive_dfprior_amd as A
from attentive_dfprior_amd import mapping, synthetic
from oracle import adfp_oracle as O
from conftest import make_cfg

rank, world, port, out, masked = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5] == '1'
DEV = torch.device('cuda:0')
if world > 1:
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:' + port, rank=rank, world_size=world)
sc = synthetic.mini_scene()
dec = A.DF(); dec.load_state_dict(O.random_state_dict(seed=3)); dec.bound = sc.bound; dec = dec.to(DEV)
for p in list(dec.low_decoder.parameters()) + list(dec.high_decoder.parameters()):
    p.requires_grad_(False)
rays = [t.to(DEV) for t in synthetic.make_ray_batch(sc, 600, seed=5)]
lo, hi = (rank * 600) // world, ((rank + 1) * 600) // world
mine = [t[lo:hi].contiguous() for t in rays]
masks = None
if masked:
    c2w = sc.default_c2w(yaw=0.7, pitch=0.1)
    masks = {k: mapping.frustum_mask(c2w, tuple(v.shape[2:]), sc.depth_image(c2w).to(DEV), sc.bound, sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy)
             for k, v in sc.c.items()}
lr = {s: dict(low=0.01, high=0.005, color=0.005, decoders=0.005, mlp=0.005) for s in ('low', 'high', 'color')}
grids = {k: v.clone().to(DEV) for k, v in sc.c.items()}
it = mapping.MapperIteration(A.Renderer(make_cfg(32, 16), None, sc), dec, grids, masks, sc.tsdf_volume.to(DEV), sc.tsdf_bnds.to(DEV), lr,
                             use_graph=False, distributed=world > 1)
losses = [float(it.step(*mine, stage)) for stage in ('low', 'high', 'color', 'color')]
if rank == 0:
    torch.save({'grids': {k: v.cpu() for k, v in grids.items()}, 'params': {n: p.detach().cpu() for n, p in dec.named_parameters()},
                'losses': losses, 'bucket_bytes': it.bucket_bytes, 'dense_bytes': 4 * it.bucket.numel()}, out)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return str(s.getsockname()[1])


def _run(world, masked, tmp):
    out = os.path.join(tmp, f'w{world}.pt')
    port = _free_port()
    code = WORKER % {'root': ROOT, 'here': HERE}
    procs = [subprocess.Popen([sys.executable, '-c', code, str(r), str(world), port, out, '1' if masked else '0'], env=dict(os.environ))
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    return torch.load(out)


@pytest.mark.parametrize('masked', [False, True])
def test_two_ranks_follow_the_single_process_iteration(masked):
    with tempfile.TemporaryDirectory() as tmp:
        one = _run(1, masked, tmp)
        two = _run(2, masked, tmp)
    for a, b in zip(two['losses'], one['losses']):                 # the all-reduced loss of the ranks = the loss of all rays
        assert abs(a - b) <= 1e-5 * abs(b), (two['losses'], one['losses'])
    for k in one['grids']:
        assert_adam_trajectory(two['grids'][k], one['grids'][k], 0.01, 4, k, max_outliers=1e-2)
    for n in one['params']:
        assert_adam_trajectory(two['params'][n], one['params'][n], 0.005, 4, n, max_outliers=8e-2)      # (see tests/test_gpu_nccl.py)
    if masked:
        assert two['bucket_bytes'] < two['dense_bytes']            # only the selected voxels (and the parameters) travelled
