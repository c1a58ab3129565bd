"""Debug build only (tools/experiments/build_att_span.sh -> tools/ab_libs/libadfp_att_span.so): the timeline of the workgroups of the
attention network's training forward (k_attention_h<1, 256>, one wave per SIMD) in the fused Mapper iteration.
    ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_att_span.so python tools/experiments/att_span.py [rays] [N_samples]"""
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
buf = (C.c_ulonglong * (16 * 1024))()
L.adfp_debug_att_span.argtypes = [C.c_void_p]
assert L.adfp_debug_att_span(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.int64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
us = lambda x: (x - t0) / 100.0
nt = a[:, 3]
print(f'{rays} rays x {ns + 16} samples: {len(a)} workgroups; tiles of wave 0: min {nt.min()} p50 {int(np.median(nt))} max {nt.max()}')
print(f'start us: min {us(a[:, 0]).min():.1f} p50 {np.median(us(a[:, 0])):.1f} max {us(a[:, 0]).max():.1f}')
w = (a[:, 1] - a[:, 0]) / 100.0
print(f'weight image in LDS after start us: min {w.min():.2f} p50 {np.median(w):.2f} max {w.max():.2f}')
for k in range(min(6, int(nt.max()))):
    m = nt > k
    d = (a[m, 5 + 2 * k] - a[m, 4 + 2 * k]) / 100.0
    print(f'tile {k} of wave 0: {int(m.sum())} workgroups, us min {d.min():.2f} p50 {np.median(d):.2f} max {d.max():.2f}')
print(f'wave 0 done us: min {us(a[:, 2]).min():.1f} p50 {np.median(us(a[:, 2])):.1f} max {us(a[:, 2]).max():.1f}')
