"""cProfile of the reference-shaped training iteration (render_batch_ray under autograd + Mapper loss + backward + torch Adam):
where the HOST time of the path an unchanged Mapper.py takes goes.   python tools/host_profile.py [rays]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import attentive_dfprior_amd as A                                    # noqa: E402
from attentive_dfprior_amd import synthetic                          # noqa: E402
from attentive_dfprior_amd.common import get_rays                    # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device('cuda:0')
scene = synthetic.Scene('room0', device=dev, grid_std_scale=20.0)
scene.c['grid_high'] = scene.c['grid_high'] * 100
dec = A.DF()
dec.load_state_dict(synthetic.seeded_state_dict(0))
dec.bound = scene.bound
dec = dec.to(dev)
for p in list(dec.low_decoder.parameters()) + list(dec.high_decoder.parameters()):
    p.requires_grad_(False)
cfg = {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': 32, 'N_surface': 16, 'N_importance': 0},
       'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}
rend = A.Renderer(cfg, None, scene)
tb = scene.tsdf_bnds.to(dev)
c2w = scene.default_c2w()
gt = scene.depth_image(c2w)
ro, rd = get_rays(scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy, c2w, dev)
pick = torch.randint(scene.H * scene.W, (n,), generator=torch.Generator().manual_seed(0)).to(dev)
ro, rd, gd = ro.reshape(-1, 3)[pick], rd.reshape(-1, 3)[pick], gt.reshape(-1)[pick]
gc = torch.rand(n, 3, device=dev)
c = {k: v.clone().requires_grad_(True) for k, v in scene.c.items()}
params = list(dec.color_decoder.parameters()) + list(dec.mlp.parameters())
opt = torch.optim.Adam([{'params': params, 'lr': 0.005}, {'params': list(c.values()), 'lr': 0.005}])


def it():
    opt.zero_grad()
    d, u, col, w = rend.render_batch_ray(c, dec, rd, ro, dev, scene.tsdf_volume, tb, 'color', gt_depth=gd)
    m = gd > 0
    loss = torch.abs(gd[m] - d[m]).sum() + 0.2 * torch.abs(gc - col).sum()
    loss.backward()
    opt.step()


for _ in range(10):
    it()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(100):
    it()
torch.cuda.synchronize()
print('ms per iteration', (time.perf_counter() - t0) * 10)
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    it()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
