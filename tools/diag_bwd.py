"""Diagnostic: the f16-split backward against the exact f32 backward on the gradient tests' second-seed case.
Modes: f32 (exact forward + backward), f16x3 (split forward + split backward), mix (split forward, exact backward: the
training state carries no masks).  Prints per-tensor max relative error against the f32 run and, for the worst tensor, the
per-row errors."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import attentive_dfprior_amd as A
from attentive_dfprior_amd import synthetic, engine
from oracle import adfp_oracle as O
from conftest import make_cfg

DEV = torch.device('cuda:0')


def mapper_loss(d, col, w, gd, gc, stage, warm):
    m = gd > 0
    loss = torch.abs(gd[m] - d[m]).sum()
    if warm:
        loss = loss + torch.abs(w - 1.0).sum()
    if stage == 'color':
        loss = loss + 0.2 * torch.abs(gc - col).sum()
    return loss


def run(mode, sc, sd, rays, stage='color'):
    os.environ['ADFP_MATH'] = 'f32' if mode == 'f32' else 'f16x3'
    orig = engine.Engine.train_state
    if mode.startswith('mix'):
        def no_masks(*args, **kwargs):            # (whatever Engine.train_state takes: round 4 added `extra` and this tool broke silently)
            os.environ['ADFP_MATH'] = 'f32'
            try:
                return orig(*args, **kwargs)
            finally:
                os.environ['ADFP_MATH'] = 'f16x3'
        engine.Engine.train_state = staticmethod(no_masks)
    try:
        dec = A.DF()
        dec.load_state_dict(sd)
        dec.bound = sc.bound
        dec = dec.to(DEV)
        rend = A.Renderer(make_cfg(48, 16), None, sc)
        c = {k: v.to(DEV).clone().requires_grad_(True) for k, v in sc.c.items()}
        ro, rd, gd, gc = [t.to(DEV) for t in rays]
        d, u, col, w = rend.render_batch_ray(c, dec, rd, ro, DEV, sc.tsdf_volume.to(DEV), sc.tsdf_bnds.to(DEV), stage, gt_depth=gd)
        loss = mapper_loss(d, col, w, gd, gc, stage, True)
        loss.backward()
        out = {k: v.grad.detach().double().cpu() for k, v in c.items() if v.grad is not None}
        out.update({n: p.grad.detach().double().cpu() for n, p in dec.named_parameters() if p.grad is not None})
        return out
    finally:
        engine.Engine.train_state = staticmethod(orig)


def main():
    sc = synthetic.mini_scene()
    sd = O.random_state_dict(seed=17)
    rays = synthetic.make_ray_batch(sc, 300, seed=8, poses=3)
    runs = {m: run(m, sc, sd, rays) for m in ('f32', 'mix', 'mix2', 'f16x3')}
    for mode, base in (('mix', 'f32'), ('f16x3', 'f32'), ('mix2', 'mix'), ('f16x3', 'mix')):
        got, ref = runs[mode], runs[base]
        worst = None
        print(f'---- {mode} vs {base}')
        for k in ref:
            scale = ref[k].abs().max().item()
            err = (got[k] - ref[k]).abs().max().item()
            rel = err / max(scale, 1e-30)
            if rel > float(os.environ.get('DIAG_TOL', '1e-4')):
                print(f'{k:44s} rel {rel:.2e}  scale {scale:.2e}')
            if worst is None or rel > worst[1]:
                worst = (k, rel)
        k = worst[0]
        print('worst', worst)
        if ref[k].dim() == 2:
            e = (got[k] - ref[k]).abs()
            print('per-row max err / tensor scale:', [f'{v:.1e}' for v in (e.max(dim=1).values / ref[k].abs().max()).tolist()])


if __name__ == '__main__':
    main()
