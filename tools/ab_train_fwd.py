"""Timing of the TRAINING forward alone (render_forward(train=True), colour stage, colour decoder + attention MLP trainable) at the
Mapper's batch sizes: ms per call by HIP events, for kernel A/B builds (ADFP_LIB_PATH) whose backward state may be incomplete
(timing-only switches ADFP_EXP_TRAIN_*): nothing here runs a backward."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import attentive_dfprior_amd as A                                    # noqa: E402
from attentive_dfprior_amd import synthetic                          # noqa: E402
from attentive_dfprior_amd.common import get_rays                    # noqa: E402

dev = torch.device('cuda:0')
rays, ns = int(sys.argv[1]) if len(sys.argv) > 1 else 5000, int(sys.argv[2]) if len(sys.argv) > 2 else 48
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
scene = synthetic.Scene('room0', device=dev, grid_std_scale=20.0)
scene.c['grid_high'] = scene.c['grid_high'] * 100
dec = A.DF()
dec.load_state_dict(synthetic.seeded_state_dict(0))
dec.bound = scene.bound
dec = dec.to(dev)
cfg = {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': ns, 'N_surface': 16, 'N_importance': 0},
       'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}
rend = A.Renderer(cfg, None, scene)
c2w = scene.default_c2w()
gt = scene.depth_image(c2w)
ro, rd = get_rays(scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy, c2w, dev)
pick = torch.randint(scene.H * scene.W, (rays,), generator=torch.Generator().manual_seed(0)).to(dev)
ro, rd, gd = ro.reshape(-1, 3)[pick].contiguous(), rd.reshape(-1, 3)[pick].contiguous(), gt.reshape(-1)[pick].contiguous()
tb = scene.tsdf_bnds.to(dev)
need = {'low': False, 'high': False, 'color': True, 'att': True}


def fwd():
    return rend._engine.render_forward(dec, scene.c, ro, rd, gd, scene.tsdf_volume, tb, scene.bound, 'color', ns, 16, train=True, need_flat=need)


for _ in range(5):
    fwd()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    fwd()
e1.record()
torch.cuda.synchronize()
print(f'training forward {rays} x {ns + 16}: {e0.elapsed_time(e1) / iters:.4f} ms per call ({os.environ.get("ADFP_LIB_PATH", "in-tree")})')
