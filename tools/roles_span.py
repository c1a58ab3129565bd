"""Debug build only (-DADFP_STAMPS_ROLES -> tools/ab_libs/libadfp_roles_span.so): when does each workgroup of k_decode_bwd_roles start,
leave its tile loop and end, by role, in the 5 000-ray x 64-sample Mapper iteration.
    ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_roles_span.so python tools/roles_span.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import attentive_dfprior_amd as A                                    # noqa: E402
from attentive_dfprior_amd import synthetic, mapping, _lib           # noqa: E402
from attentive_dfprior_amd.common import get_rays                    # noqa: E402

dev = torch.device('cuda:0')
rays, ns = 5000, 48
scene = synthetic.Scene('room0', device=dev, grid_std_scale=20.0)
scene.c['grid_high'] = scene.c['grid_high'] * 100
dec = A.DF(); dec.load_state_dict(synthetic.seeded_state_dict(0)); dec.bound = scene.bound; dec = dec.to(dev)
cfg = {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': ns, 'N_surface': 16, 'N_importance': 0}, 'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}
rend = A.Renderer(cfg, None, scene)
c2w = scene.default_c2w()
gt = scene.depth_image(c2w)
ro, rd = get_rays(scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy, c2w, dev)
pick = torch.randint(scene.H * scene.W, (rays,), generator=torch.Generator().manual_seed(0)).to(dev)
ro, rd, gd = ro.reshape(-1, 3)[pick], rd.reshape(-1, 3)[pick], gt.reshape(-1)[pick]
gc = torch.rand(rays, 3, device=dev)
masks = {k: mapping.frustum_mask(c2w, tuple(v.shape[2:]), gt, scene.bound, scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy) for k, v in scene.c.items()}
lr = {'color': dict(low=0.005, high=0.005, color=0.005, decoders=0.005, mlp=0.005)}
it = mapping.MapperIteration(rend, dec, {k: v.clone() for k, v in scene.c.items()}, masks, scene.tsdf_volume, scene.tsdf_bnds.to(dev), lr, use_graph=False)
for _ in range(5):
    it.step(ro, rd, gd, gc, 'color')
torch.cuda.synchronize()
L = _lib.lib()
buf = (C.c_ulonglong * 1024)()
L.adfp_debug_roles_span.argtypes = [C.c_void_p]
assert L.adfp_debug_roles_span(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 4).astype(np.int64)
t0 = a[:, 1].min()
for r, name in enumerate('PHC'):
    m = a[:, 0] == r
    s, l, e = (a[m, 1] - t0) / 100.0, (a[m, 2] - t0) / 100.0, (a[m, 3] - t0) / 100.0
    print(f'role {name}: {int(m.sum())} workgroups; start us min {s.min():.1f} max {s.max():.1f}; tile loop done us min {l.min():.1f} p50 {np.median(l):.1f} max {l.max():.1f}; '
          f'end us min {e.min():.1f} p50 {np.median(e):.1f} max {e.max():.1f}')
