"""Diagnostic: how the sample points of one Mapper iteration distribute over the 4-cell bins of the finest grid (the work
distribution of k_scatter_sorted: points per 4-cell bin of a grid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import attentive_dfprior_amd as A                                    # noqa: E402
from attentive_dfprior_amd import synthetic                          # noqa: E402
from attentive_dfprior_amd.common import get_rays                    # noqa: E402

dev = torch.device('cuda:0')
scene = synthetic.Scene('room0', device=dev, grid_std_scale=20.0)
dec = A.DF(); dec.load_state_dict(synthetic.seeded_state_dict(0)); dec.bound = scene.bound; dec = dec.to(dev)
cfg = {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': 48, 'N_surface': 16, 'N_importance': 0},
       'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}
rend = A.Renderer(cfg, None, scene)
c2w = scene.default_c2w()
gt = scene.depth_image(c2w)
ro, rd = get_rays(scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy, c2w, dev)
pick = torch.randint(scene.H * scene.W, (5000,), generator=torch.Generator().manual_seed(0)).to(dev)
ro, rd, gd = ro.reshape(-1, 3)[pick], rd.reshape(-1, 3)[pick], gt.reshape(-1)[pick]
d, u, col, w, aux = rend._engine.render_forward(dec, scene.c, ro, rd, gd, scene.tsdf_volume, scene.tsdf_bnds.to(dev), rend.bound, 'color', 48, 16,
                                                want_aux=True)
z = aux['z_vals']
pts = ro[:, None, :].double() + rd[:, None, :].double() * z[..., None]
b = scene.bound.to(dev).double()
pn = ((pts - b[:, 0]) / (b[:, 1] - b[:, 0]) * 2 - 1).float().reshape(-1, 3)
print('points', pn.shape[0], 'outside bound', float(((pn.abs() > 1).any(dim=1)).float().mean()), 'nan', int(torch.isnan(pn).any(dim=1).sum()))
for key in ('grid_low', 'grid_color'):
    Z, Y, X = scene.c[key].shape[2:]
    dims = torch.tensor([X, Y, Z], device=dev)
    c = ((pn + 1) / 2 * (dims - 1)).clamp(min=0)
    c = torch.minimum(c, (dims - 1).float())
    i0 = c.floor().long()
    i0 = torch.minimum(i0, dims - 2)
    bins = i0 >> 2
    nb = ((dims - 2) >> 2) + 1
    lin = (bins[:, 2] * nb[1] + bins[:, 1]) * nb[0] + bins[:, 0]
    cnt = torch.bincount(lin, minlength=int(nb.prod()))
    print(key, 'dims', (X, Y, Z), 'bins', int(nb.prod()), 'non-empty', int((cnt > 0).sum()), 'max', int(cnt.max()), 'top5', sorted(cnt.tolist())[-5:],
          'median of non-empty', float(cnt[cnt > 0].float().median()))
