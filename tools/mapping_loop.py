"""BASELINE.json configs[2]: Replica office0-sized scene, 200-frame mapping loop, 5 000 rays per iteration
(training stability).  Follows the structure of Mapper.optimize_map (reference src/Mapper.py:374-473): a
fresh Adam per frame over {decoders, mlp, low, high, color} parameter groups, the low -> high -> color stage
schedule by iteration ratio with the per-stage learning rates of configs/df_prior.yaml:65-83, rays drawn from
the current frame and 4 keyframes with get_samples, the bbox pre-filter, the Mapper loss, frustum feature
selection (src/Mapper.py:330-361): only the grid points in the current frustum are optimised.  The frustum
mask, the pre-filter and the masked Adam run in libadfp.so (attentive_dfprior_amd.mapping / .common).
Synthetic poses on a circle; ITERS iterations per frame instead of 60 to keep the run short.  Not part of
the driver contract.

  python tools/mapping_loop.py [--frames 200] [--iters 10] [--rays 5000]
"""
import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import attentive_dfprior_amd as A                                   # noqa: E402
from attentive_dfprior_amd import synthetic, common, mapping        # noqa: E402

STAGE_LR = {'low': dict(mlp=0.0, dec=0.0, low=0.1, high=0.0, color=0.0),
            'high': dict(mlp=0.005, dec=0.0, low=0.005, high=0.005, color=0.0),
            'color': dict(mlp=0.005, dec=0.005, low=0.005, high=0.005, color=0.005)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=200)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--rays', type=int, default=5000)
    ap.add_argument('--scene', default='office0')
    ap.add_argument('--profile', action='store_true', help='synchronise after every section and print where the time goes')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    sc = synthetic.Scene(args.scene, device=dev)
    dec = A.DF()
    dec.load_state_dict(synthetic.seeded_state_dict(0))
    dec.bound = sc.bound
    dec = dec.to(dev)
    cfg = {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': 48, 'N_surface': 16, 'N_importance': 0},
           'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}
    rend = A.Renderer(cfg, None, sc)
    tsdf_bnds = sc.tsdf_bnds.to(dev)
    bound = sc.bound.to(dev)
    H, W, fx, fy, cx, cy = sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy
    gen = torch.Generator().manual_seed(0)
    target_color = torch.rand(H, W, 3, generator=gen).to(dev)
    c = {k: v.clone() for k, v in sc.c.items()}
    keyframes = []
    hist = []
    torch.manual_seed(0)
    sect = {}

    def tick(name, t0):
        if args.profile:
            torch.cuda.synchronize()
            sect[name] = sect.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()

    t_start = time.perf_counter()
    n_iter = 0
    for f in range(args.frames):
        tk = time.perf_counter()
        ang = 2 * math.pi * f / args.frames
        c2w = sc.default_c2w(offset=(1.5 * math.cos(ang), 1.5 * math.sin(ang), 0.2 * math.sin(3 * ang)), yaw=ang, pitch=-0.1)
        depth = sc.depth_image(c2w)
        if f % 5 == 0:
            keyframes.append((c2w, depth))
        frames = [(c2w, depth)] + keyframes[-4:]
        grids = {k: v.detach().requires_grad_(True) for k, v in c.items()}
        masks = {k: mapping.frustum_mask(c2w, tuple(v.shape[2:]), depth, sc.bound, H, W, fx, fy, cx, cy) for k, v in grids.items()}
        opt_grids = mapping.MaskedGridAdam(grids, masks)
        opt = torch.optim.Adam([{'params': list(dec.color_decoder.parameters()), 'lr': 0},      # fix_high: True
                                {'params': list(dec.mlp.parameters()), 'lr': 0}])
        first = None
        tk = tick('frame setup (pose, depth image, optimizer)', tk)
        for it in range(args.iters):
            stage = 'low' if it <= int(args.iters * 0.4) else ('high' if it <= int(args.iters * 0.6) else 'color')
            lr = STAGE_LR[stage]
            for g, key in zip(opt.param_groups, ('dec', 'mlp')):
                g['lr'] = lr[key]
            opt.zero_grad()
            opt_grids.zero_grad()
            ros, rds, gds, gcs = [], [], [], []
            for kc2w, kdepth in frames:
                ro, rd, gd, gc = common.get_samples(0, H, 0, W, args.rays // len(frames), H, W, fx, fy, cx, cy, kc2w,
                                                    kdepth, target_color, dev)
                ros.append(ro.float()); rds.append(rd.float()); gds.append(gd.float()); gcs.append(gc.float())
            ro, rd, gd, gc = torch.cat(ros), torch.cat(rds), torch.cat(gds), torch.cat(gcs)
            tk = tick('get_samples x frames', tk)
            ro, rd, gd, gc = common.filter_rays_in_bound(ro, rd, gd, gc, bound)     # src/Mapper.py:439-449
            tk = tick('bbox pre-filter', tk)
            d, u, col, w = rend.render_batch_ray(grids, dec, rd, ro, dev, sc.tsdf_volume, tsdf_bnds, stage, gd)
            tk = tick('render forward', tk)
            m = gd > 0
            loss = torch.abs(gd[m] - d[m]).sum()
            if stage == 'color':
                loss = loss + 0.2 * torch.abs(gc - col).sum()
            tk = tick('loss', tk)
            loss.backward()
            tk = tick('backward', tk)
            opt.step()
            opt_grids.step({'grid_low': lr['low'], 'grid_high': lr['high'], 'grid_color': lr['color']})
            tk = tick('Adam', tk)
            n_iter += 1
            if first is None:
                first = float(loss) / max(1, int(m.sum()))
            tk = tick('bookkeeping', tk)
        c = {k: v.detach() for k, v in grids.items()}
        hist.append((first, float(loss) / max(1, int(m.sum()))))
        if not math.isfinite(hist[-1][1]):
            raise SystemExit(f'non-finite loss at frame {f}')
    torch.cuda.synchronize()
    dt = time.perf_counter() - t_start
    q = len(hist) // 4
    mean = lambda xs: sum(xs) / len(xs)
    print(json.dumps({'config': f'{args.scene} mapping loop', 'frames': args.frames, 'iters_per_frame': args.iters,
                      'rays_per_iter': args.rays, 'iterations': n_iter, 'seconds': dt, 'ms_per_iteration': dt / n_iter * 1e3,
                      'depth_loss_per_ray_first_quarter': mean([h[1] for h in hist[:q]]),
                      'depth_loss_per_ray_last_quarter': mean([h[1] for h in hist[-q:]]),
                      'sections_ms_per_iteration': {k: v / n_iter * 1e3 for k, v in sect.items()},
                      'all_finite': True, 'grid_absmax': {k: float(v.abs().max()) for k, v in c.items()}}))


if __name__ == '__main__':
    main()
