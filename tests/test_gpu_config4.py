"""GPU: BASELINE.json configs[3] -- ScanNet scene0050_00 (bounds configs/ScanNet/scene0050.yaml:3, camera 640x480
cropped by crop_edge 10 to 620x460, configs/ScanNet/scannet.yaml:24-32), the frame's rays split into 8 contiguous
shards, one per GPU, with the FULL batch's max(gt_depth) handed to every shard (src/utils/Renderer.py:159, :195).
One GPU plays every rank in turn: the concatenated shard renders must equal the unsharded render bit for bit, a
shard agrees with the oracle, and the sum of the shards' Mapper-loss gradients equals the unsharded gradient (what
the RCCL all-reduce of attentive_dfprior_amd.dist produces; the collective itself runs in test_dist_gloo.py under
gloo and in test_gpu_nccl.py under RCCL)."""
import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import synthetic, dist as adist
from attentive_dfprior_amd.common import get_rays
from oracle import adfp_oracle as O
from conftest import make_cfg, assert_close, assert_close_scale

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
WORLD = 8


@pytest.fixture(scope='module')
def scan():
    # scannet.yaml: H 480, W 640, fx 577.590698, fy 578.729797, cx 318.905426, cy 242.683609, crop_edge 10
    sc = synthetic.Scene('scene0050', H=460, W=620, fx=577.590698, fy=578.729797, cx=318.905426 - 10, cy=242.683609 - 10,
                         device=DEV, grid_std_scale=20.0, inset=0.4)
    sc.c['grid_high'] = sc.c['grid_high'] * 100
    assert tuple(sc.c['grid_high'].shape[2:]) == (21, 29, 41) and tuple(sc.tsdf_volume.shape[2:]) == (226, 308, 431)
    sd = synthetic.seeded_state_dict(0)
    dec = A.DF()
    dec.load_state_dict(sd)
    dec.bound = sc.bound
    dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(48, 16), None, sc, ray_batch_size=10 ** 9)
    c2w = sc.default_c2w(yaw=0.4, pitch=-0.15)
    gd = sc.depth_image(c2w).reshape(-1)
    ro, rd = get_rays(sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, c2w, DEV)
    return sc, sd, dec, rend, ro.reshape(-1, 3), rd.reshape(-1, 3), gd


def test_sharded_frame_equals_unsharded_bit_for_bit(scan):
    sc, sd, dec, rend, ro, rd, gd = scan
    n = ro.shape[0]
    assert n == 620 * 460
    tb = sc.tsdf_bnds.to(DEV)
    with torch.no_grad():
        full = rend.render_batch_ray(sc.c, dec, rd, ro, DEV, sc.tsdf_volume, tb, 'color', gd)
        dmax = gd.max().reshape(1)
        parts = []
        for r in range(WORLD):
            lo, hi = adist.shard_range(n, r, WORLD)
            parts.append(rend.render_batch_ray(sc.c, dec, rd[lo:hi], ro[lo:hi], DEV, sc.tsdf_volume, tb, 'color', gd[lo:hi],
                                               depth_max=dmax))
    for k in range(4):
        assert torch.equal(torch.cat([p[k] for p in parts]), full[k]), f'output {k} differs between sharded and unsharded'
    # without the full-batch max a shard clamps `far` differently: the knob matters
    lo, hi = adist.shard_range(n, 0, WORLD)
    if float(gd[lo:hi].max()) < float(gd.max()):
        with torch.no_grad():
            own = rend.render_batch_ray(sc.c, dec, rd[lo:hi], ro[lo:hi], DEV, sc.tsdf_volume, tb, 'color', gd[lo:hi])
        assert not torch.equal(own[0], full[0][lo:hi])


def test_one_rank_shard_vs_oracle(scan):
    sc, sd, dec, rend, ro, rd, gd = scan
    n = ro.shape[0]
    lo, hi = adist.shard_range(n, 5, WORLD)
    dmax = gd.max().reshape(1)
    with torch.no_grad():
        d, u, c, w = rend.render_batch_ray(sc.c, dec, rd[lo:hi], ro[lo:hi], DEV, sc.tsdf_volume, sc.tsdf_bnds.to(DEV), 'color',
                                           gd[lo:hi], depth_max=dmax)
    pick = torch.arange(0, hi - lo, 71, device=DEV)
    cpu = {k: v.cpu() for k, v in sc.c.items()}
    od, ou, oc, ow = O.render_batch_ray(sd, cpu, rd[lo:hi][pick].cpu(), ro[lo:hi][pick].cpu(), sc.tsdf_volume.cpu(), sc.tsdf_bnds,
                                        sc.bound, 'color', gd[lo:hi][pick].cpu(), 48, 16, depth_max=dmax.cpu())
    assert int(((w[pick].cpu() == 1) != (ow == 1)).sum()) == 0
    assert_close(d[pick], od, 1e-4, 'depth')
    assert_close(c[pick], oc, 1e-4, 'colour')
    assert_close(u[pick], ou, 1e-4, 'uncertainty')
    assert_close(w[pick], ow, 1e-4, 'attention weight')


def test_shard_gradients_sum_to_the_unsharded_gradient(scan):
    """What the all-reduce (SUM) of dist.allreduce_grads yields: the Mapper losses are plain sums over rays
    (src/Mapper.py:457-469), so the shard gradients add up to the single-GPU gradient with no rescaling."""
    sc, sd, dec, rend, ro, rd, gd = scan
    g = torch.Generator().manual_seed(3)
    pick = torch.randint(ro.shape[0], (5000,), generator=g).to(DEV)            # scannet.yaml: mapping.pixels 5000
    ro, rd, gd = ro[pick].contiguous(), rd[pick].contiguous(), gd[pick].contiguous()
    gc = torch.rand(5000, 3, generator=g).to(DEV)
    tb = sc.tsdf_bnds.to(DEV)
    for p in list(dec.low_decoder.parameters()) + list(dec.high_decoder.parameters()):
        p.requires_grad_(False)
    params = [p for p in dec.parameters() if p.requires_grad]

    def grads(lo, hi, dmax):
        c = {k: v.detach().clone().requires_grad_(True) for k, v in sc.c.items()}
        for p in params:
            p.grad = None
        d, u, col, w = rend.render_batch_ray(c, dec, rd[lo:hi], ro[lo:hi], DEV, sc.tsdf_volume, tb, 'color', gd[lo:hi], depth_max=dmax)
        m = gd[lo:hi] > 0
        (torch.abs(gd[lo:hi][m] - d[m]).sum() + 0.2 * torch.abs(gc[lo:hi] - col).sum()).backward()
        out = [c[k].grad.clone() for k in ('grid_low', 'grid_high', 'grid_color')]
        return out + [p.grad.clone() for p in params]

    dmax = gd.max().reshape(1)
    full = grads(0, 5000, dmax)
    acc = None
    for r in range(WORLD):
        lo, hi = adist.shard_range(5000, r, WORLD)
        part = grads(lo, hi, dmax)
        acc = part if acc is None else [a + b for a, b in zip(acc, part)]
    bucket_bytes = sum(t.numel() for t in full) * 4
    assert 6.0e6 < bucket_bytes < 8.0e6                                        # scene0050 ~ 6.8 MB
    for a, b in zip(acc, full):
        assert_close_scale(a, b, 2e-5, 'sum of shard gradients vs unsharded gradient')      # sums in a different order: held to the tensor's scale
    for p in dec.parameters():
        p.grad = None
        p.requires_grad_(True)
