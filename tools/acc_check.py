import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import attentive_dfprior_amd as A
from attentive_dfprior_amd import synthetic
from oracle import adfp_oracle as O
from conftest import make_cfg
DEV = 'cuda:0'
sc = synthetic.mini_scene(); sd = O.random_state_dict(seed=17)
rays = synthetic.make_ray_batch(sc, 300, seed=8, poses=3)
ro, rd, gd, gc = [t.to(DEV) for t in rays]
outs = {}
for mode in ('f32', 'f16x3'):
    os.environ['ADFP_MATH'] = mode
    dec = A.DF(); dec.load_state_dict(sd); dec.bound = sc.bound; dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(48, 16), None, sc)
    with torch.no_grad():
        d, u, c, w, aux = rend._engine.render_forward(dec, {k: v.to(DEV) for k, v in sc.c.items()}, ro, rd, gd, sc.tsdf_volume.to(DEV), sc.tsdf_bnds.to(DEV), sc.bound, 'color', 48, 16, want_aux=True)
    outs[mode] = aux['raw'].double().cpu()
a, b = outs['f16x3'], outs['f32']
fin = torch.isfinite(b) & (b != 100)
for ch in range(4):
    e = (a[..., ch] - b[..., ch])[fin[..., ch]].abs()
    print('channel', ch, 'max abs diff f16x3 vs f32', float(e.max()), 'rms', float((e ** 2).mean().sqrt()), 'scale', float(b[..., ch][fin[..., ch]].abs().max()))
