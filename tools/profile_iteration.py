"""One fused Mapper iteration (mapping.MapperIteration, eager kernel sequence) in a loop, for `rocprofv3 --kernel-trace --stats`:
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_iter -- python3 tools/profile_iteration.py --rays 1000 --samples 32"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import attentive_dfprior_amd as A                                    # noqa: E402
from attentive_dfprior_amd import synthetic, mapping                 # noqa: E402
from attentive_dfprior_amd.common import get_rays                    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--rays', type=int, default=1000)
ap.add_argument('--samples', type=int, default=32)
ap.add_argument('--iters', type=int, default=50)
ap.add_argument('--graph', action='store_true')
ap.add_argument('--scene', default='room0')
ap.add_argument('--stage', default='color')
ap.add_argument('--masked', action='store_true', help='frustum-masked grids (as the Mapper runs) instead of whole-grid Adam')
args = ap.parse_args()
dev = torch.device('cuda:0')
scene = synthetic.Scene(args.scene, device=dev, grid_std_scale=20.0)
scene.c['grid_high'] = scene.c['grid_high'] * 100
dec = A.DF()
dec.load_state_dict(synthetic.seeded_state_dict(0))
dec.bound = scene.bound
dec = dec.to(dev)
cfg = {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': args.samples, 'N_surface': 16, 'N_importance': 0},
       'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}
rend = A.Renderer(cfg, None, scene)
c2w = scene.default_c2w()
gt = scene.depth_image(c2w)
ro, rd = get_rays(scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy, c2w, dev)
pick = torch.randint(scene.H * scene.W, (args.rays,), generator=torch.Generator().manual_seed(0)).to(dev)
ro, rd, gd = ro.reshape(-1, 3)[pick], rd.reshape(-1, 3)[pick], gt.reshape(-1)[pick]
gc = torch.rand(args.rays, 3, device=dev)
masks = None
if args.masked:
    masks = {k: mapping.frustum_mask(c2w, tuple(v.shape[2:]), gt, scene.bound, scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy)
             for k, v in scene.c.items()}
lr = {'color': dict(low=0.005, high=0.005, color=0.005, decoders=0.005, mlp=0.005), 'high': dict(low=0.005, high=0.005, color=0.0, decoders=0.0, mlp=0.005),
      'low': dict(low=0.1, high=0.0, color=0.0, decoders=0.0, mlp=0.0)}
it = mapping.MapperIteration(rend, dec, {k: v.clone() for k, v in scene.c.items()}, masks, scene.tsdf_volume, scene.tsdf_bnds.to(dev), lr,
                             use_graph=args.graph)
import time
for _ in range(3):
    it.step(ro, rd, gd, gc, args.stage)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.iters):
    it.step(ro, rd, gd, gc, args.stage)
torch.cuda.synchronize()
print('ms per iteration', (time.perf_counter() - t0) / args.iters * 1e3)
