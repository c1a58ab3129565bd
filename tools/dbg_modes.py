import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mode = os.environ.get('ADFP_MATH', 'f16x3')
import attentive_dfprior_amd as A
from attentive_dfprior_amd import synthetic
dev = 'cuda:0'
sc = synthetic.Scene('room0', device=dev, grid_std_scale=20.0)
sc.c['grid_high'] = sc.c['grid_high'] * 100
sd = synthetic.seeded_state_dict(0)
dec = A.DF(); dec.load_state_dict(sd); dec.bound = sc.bound; dec = dec.to(dev)
cfg = {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': 48, 'N_surface': 16, 'N_importance': 0}, 'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}
rend = A.Renderer(cfg, None, sc)
c2w = sc.default_c2w(); gd = sc.depth_image(c2w)
from attentive_dfprior_amd.common import get_rays
ro, rd = get_rays(sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, c2w, dev)
ro, rd, g = ro.reshape(-1, 3)[:100000].contiguous(), rd.reshape(-1, 3)[:100000].contiguous(), gd.reshape(-1)[:100000].contiguous()
with torch.no_grad():
    d, u, c, w, aux = rend._engine.render_forward(dec, sc.c, ro, rd, g, sc.tsdf_volume, sc.tsdf_bnds.to(dev), sc.bound, 'color', 48, 16, want_aux=True)
torch.save({'d': d.cpu(), 'c': c.cpu(), 'raw': aux['raw'].cpu()}, f'/tmp/out_{mode}.pt')
print(mode, 'raw occ absmax', float(aux['raw'][..., 3][aux['raw'][..., 3] < 99].abs().max()), 'rgb absmax', float(aux['raw'][..., :3].abs().max()))
if os.path.exists('/tmp/out_f32.pt') and os.path.exists('/tmp/out_f16x3.pt'):
    a, b = torch.load('/tmp/out_f32.pt'), torch.load('/tmp/out_f16x3.pt')
    for k in ('d', 'c', 'raw'):
        diff = (a[k].double() - b[k].double()).abs()
        print(k, 'max abs diff f32 vs f16x3', float(diff.max()), 'scale', float(a[k].abs().max()), 'n>1e-4*scale', int((diff > 1e-4 * a[k].abs().max()).sum()))
    diff = (a['raw'][..., :3] - b['raw'][..., :3]).abs().max(-1)[0]
    idx = torch.nonzero(diff > 1e-3)
    print('bad samples', idx[:10].tolist())
    for r, s in idx[:5].tolist():
        print(r, s, a['raw'][r, s].tolist(), b['raw'][r, s].tolist())
