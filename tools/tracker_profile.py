"""Profile target: N fused Tracker iterations (tracking.TrackerIteration, eager -- every kernel visible to rocprofv3).
`rocprofv3 --kernel-trace --stats -d out -- python3 tools/tracker_profile.py [rays] [iters]`"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import attentive_dfprior_amd as A  # noqa: E402
from attentive_dfprior_amd import synthetic  # noqa: E402
from attentive_dfprior_amd.tracking import TrackerIteration  # noqa: E402
import bench  # noqa: E402
import bench_extra as BX  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device('cuda:0')
scene, sd, dec = bench.build_scene(A, synthetic, 'room0', dev)
for p in dec.parameters():
    p.requires_grad_(False)
rend = A.Renderer(BX._cfg(48, 16), None, scene)
c2w = scene.default_c2w()
depth = scene.depth_image(c2w)
color = torch.rand((scene.H, scene.W, 3), generator=torch.Generator().manual_seed(0)).to(dev)
it = TrackerIteration(rend, dec, scene.c, scene.tsdf_volume, scene.tsdf_bnds.to(dev), scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy,
                      20, 20, use_graph=False)
cam = BX._tensor_from_c2w(c2w).to(dev)
cam[4:] += 0.01
it.new_frame(cam, depth, color)
for _ in range(iters):
    it.step(n)
torch.cuda.synchronize()
print('loss', it.loss.item(), 'best', it.best_loss.item())
