"""Debug build only (-DADFP_STAMPS_G -> tools/ab_libs/libadfp_phg.so): wave-cycles per phase of a tile in k_decode_high_g and in the
two networks of k_decode_lc16, one 100 000-ray batch (tools/ab_stage.py's workload).
    ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_phg.so python tools/phase_g.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import attentive_dfprior_amd as A                                    # noqa: E402
from attentive_dfprior_amd import synthetic, _lib                    # noqa: E402
from attentive_dfprior_amd.common import get_rays                    # noqa: E402

dev = torch.device('cuda:0')
L = _lib.lib()
scene = synthetic.Scene('room0', device=dev, grid_std_scale=20.0)
scene.c['grid_high'] = scene.c['grid_high'] * 100
dec = A.DF(); dec.load_state_dict(synthetic.seeded_state_dict(0)); dec.bound = scene.bound; dec = dec.to(dev)
cfg = {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': 48, 'N_surface': 16, 'N_importance': 0}, 'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}
rend = A.Renderer(cfg, None, scene)
eng = rend._engine
tb = scene.tsdf_bnds.to(dev)
c2w = scene.default_c2w()
gd_img = scene.depth_image(c2w)
ro, rd = get_rays(scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy, c2w, dev)
N, S = 100000, 64
ro, rd, gd = ro.reshape(-1, 3)[:N].contiguous(), rd.reshape(-1, 3)[:N].contiguous(), gd_img.reshape(-1)[:N].contiguous()
def render():
    with torch.no_grad():
        return eng.render_forward(dec, scene.c, ro, rd, gd, scene.tsdf_volume, tb, scene.bound, 'color', 48, 16)
for _ in range(3):
    out = render()
torch.cuda.synchronize()
band = float((out[3] != 1).float().mean())
ph = (C.c_ulonglong * 24)()
L.adfp_debug_phases_g.argtypes = [C.c_void_p, C.c_int]
L.adfp_debug_phases_g(ph, 1)
render(); torch.cuda.synchronize()
L.adfp_debug_phases_g(ph, 1)
names = ['claim + point', 'gather + exchange + split c', 'Fourier', 'five layers', 'output layer', 'stores']
tiles_lc = N * S / 32
tiles_hi = band * N * S / 32
print('in-band fraction %.4f' % band)
print('tiles processed: k_decode_high_g', ph[6], 'expected', int(band * N * S) // 32 + 1, '| k_decode_lc16', ph[14], 'expected', N * S // 32)
print('k_decode_high_g  wave-cycles per tile:', {n: round(ph[k] / tiles_hi) for k, n in enumerate(names)}, 'sum', round(sum(ph[:6]) / tiles_hi))
print('k_decode_lc16 low    :', {n: round(ph[8 + k] / tiles_lc) for k, n in enumerate(names[:5])}, 'sum', round(sum(ph[8:13]) / tiles_lc))
print('k_decode_lc16 colour :', {n: round(ph[16 + k] / tiles_lc) for k, n in enumerate(names)}, 'sum', round(sum(ph[16:22]) / tiles_lc))
