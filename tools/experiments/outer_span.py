"""Debug build only (tools/experiments/build_outer_span.sh -> tools/ab_libs/libadfp_outer_span.so): the timeline of k_outer_h's workgroups
(the attention network's weight gradients) in the fused Mapper iteration -- start, first tile, each tile, end of the tile loop, end.
    ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_outer_span.so python tools/experiments/outer_span.py [rays] [N_samples]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import attentive_dfprior_amd as A                                    # noqa: E402
from attentive_dfprior_amd import synthetic, mapping, _lib           # noqa: E402
from attentive_dfprior_amd.common import get_rays                    # noqa: E402

dev = torch.device('cuda:0')
rays = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 48
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
buf = (C.c_ulonglong * (32 * 256))()
L.adfp_debug_outer_span.argtypes = [C.c_void_p]
assert L.adfp_debug_outer_span(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 32).astype(np.int64)
a = a[a[:, 0] > 0]
a = a[a[:, 3] > 0]                                  # workgroups that had rows
t0 = a[:, 0].min()
us = lambda x: (x - t0) / 100.0
nt = a[:, 24]
print(f'{rays} rays x {ns + 16} samples: {len(a)} workgroups with rows; tiles per workgroup min {nt.min()} p50 {int(np.median(nt))} max {nt.max()} (sum {nt.sum()} = {nt.sum() * 16} rows)')
print(f'start us: min {us(a[:, 0]).min():.1f} max {us(a[:, 0]).max():.1f}')
print(f'first fetch issued after start us: p50 {np.median(a[:, 1] - a[:, 0]) / 100:.2f} max {(a[:, 1] - a[:, 0]).max() / 100:.2f}')
print(f'first tile done after start us: min {((a[:, 4] - a[:, 0]) / 100).min():.2f} p50 {np.median(a[:, 4] - a[:, 0]) / 100:.2f} max {((a[:, 4] - a[:, 0]) / 100).max():.2f}')
for k in range(1, min(20, int(nt.max()))):
    m = nt > k
    d = (a[m, 4 + k] - a[m, 3 + k]) / 100.0
    print(f'tile {k}: {int(m.sum())} workgroups, us per tile min {d.min():.2f} p50 {np.median(d):.2f} max {d.max():.2f}')
print(f'tile loop done us: min {us(a[:, 2]).min():.1f} p50 {np.median(us(a[:, 2])):.1f} max {us(a[:, 2]).max():.1f}')
e = (a[:, 3] - a[:, 2]) / 100.0
print(f'write-out us: min {e.min():.2f} p50 {np.median(e):.2f} max {e.max():.2f}')
print(f'end us: min {us(a[:, 3]).min():.1f} p50 {np.median(us(a[:, 3])):.1f} max {us(a[:, 3]).max():.1f}')
