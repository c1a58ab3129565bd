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

# per-wave wall-clock span of the last k_decode_lc16 launch (256 workgroups x 12 waves)
import numpy as np
nw = 256 * 12
buf = (C.c_ulonglong * (2 * nw))()
L.adfp_debug_wave_span_g.argtypes = [C.c_void_p, C.c_int]
assert L.adfp_debug_wave_span_g(buf, nw) == 0
raw_ = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 2)
ntile_ = (raw_[:, 0] >> np.uint64(48)).astype(np.int64)
a = np.stack([(raw_[:, 0] & np.uint64((1 << 48) - 1)).astype(np.int64), raw_[:, 1].astype(np.int64)], 1)
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0          # us
wg = en.reshape(256, 12).max(1)
print('k_decode_lc16 waves: start us min %.1f max %.1f | end us min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f' % (
    st.min(), st.max(), en.min(), np.percentile(en, 10), np.median(en), np.percentile(en, 90), en.max()))
print('per-workgroup end us: min %.1f p50 %.1f max %.1f; mean idle tail per wave %.1f us = %.2f %% of the launch' % (
    wg.min(), np.median(wg), wg.max(), (en.max() - en).mean(), 100 * (en.max() - en).mean() / en.max()))
print('per-XCD mean workgroup end us:', wg.reshape(32, 8).mean(0).round(1).tolist())
for b in (0, 100, 255):
    print('workgroup', b, 'wave ends us:', en.reshape(256, 12)[b].round(0).tolist(), 'tiles per wave:', ntile_.reshape(256, 12)[b].tolist())
print('within-workgroup spread (max - min of wave ends) us: mean %.1f max %.1f' % ((en.reshape(256, 12).max(1) - en.reshape(256, 12).min(1)).mean(), (en.reshape(256, 12).max(1) - en.reshape(256, 12).min(1)).max()))
