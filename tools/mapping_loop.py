"""BASELINE.json configs[2]: Replica office0-sized scene, 200-frame mapping loop, 5 000 rays per iteration
(training stability).  Follows the structure of Mapper.optimize_map (reference src/Mapper.py:374-473): a
fresh Adam per frame over {decoders, mlp, low, high, color} parameter groups, the low -> high -> color stage
schedule by iteration ratio with the per-stage learning rates of configs/df_prior.yaml:65-83 (times
lr_first_factor on the first frame, :60), rays drawn from the current frame and the window's keyframes with
get_samples, the bbox pre-filter, the Mapper loss (with its warm-up term on the first two frames), frustum
feature selection (src/Mapper.py:330-361): only the grid points in the current frustum are optimised.  The
frustum mask, the pre-filter, the loss and the masked Adam run in libadfp.so (attentive_dfprior_amd.mapping /
.common).  Synthetic poses on a circle.  Not part of the driver contract; tests/test_gpu_config3.py runs it.

  python tools/mapping_loop.py [--frames 200] [--iters 60] [--iters-first 300] [--rays 5000]
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

# configs/df_prior.yaml:65-83 (mapping.stage.*); configs/Replica/replica.yaml's top-level `stage:` block does not
# override it (SURVEY.md section 5 "Trap")
STAGE_LR = {'low': dict(mlp=0.0, dec=0.0, low=0.1, high=0.0, color=0.0),
            'high': dict(mlp=0.005, dec=0.0, low=0.005, high=0.005, color=0.0),
            'color': dict(mlp=0.005, dec=0.005, low=0.005, high=0.005, color=0.005)}
LOW_ITER_RATIO, HIGH_ITER_RATIO = 0.4, 0.6          # configs/df_prior.yaml:41-42
W_COLOR_LOSS = 0.2                                  # :56
LR_FIRST_FACTOR = 5                                 # :60
CFG = {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': 48, 'N_surface': 16, 'N_importance': 0},
       'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}


def circle_pose(sc, f, frames, radius=1.5):
    ang = 2 * math.pi * f / frames
    return sc.default_c2w(offset=(radius * math.cos(ang), radius * math.sin(ang), 0.2 * math.sin(3 * ang)), yaw=ang, pitch=-0.1)


def stage_of(it, n_iters):
    """src/Mapper.py:390-395"""
    if it <= int(n_iters * LOW_ITER_RATIO):
        return 'low'
    return 'high' if it <= int(n_iters * HIGH_ITER_RATIO) else 'color'


def mapper_loss(stage, it, n_iters, frame_idx, gd, gc, depth, color, weight):
    """src/Mapper.py:457-469, written with torch ops (the tool's reference form; the product also offers the fused
    device-side loss through Renderer-level helpers)."""
    m = gd > 0
    loss = torch.abs(gd[m] - depth[m]).sum()
    low_end = int(n_iters * LOW_ITER_RATIO)
    if low_end < it <= low_end + 5 and frame_idx <= 1:
        loss = loss + torch.abs(weight - 1.0).sum()
    if stage == 'color':
        loss = loss + W_COLOR_LOSS * torch.abs(gc - color).sum()
    return loss, m


class MappingRun(object):
    def __init__(self, scene='office0', rays=5000, total_frames=200, seed=0, device='cuda:0', window=5, keyframe_every=5, fused=False):
        self.fused, self.iteration = fused, None
        self.dev = torch.device(device)
        self.sc = synthetic.Scene(scene, device=self.dev)
        self.dec = A.DF()
        self.dec.load_state_dict(synthetic.seeded_state_dict(seed))
        self.dec.bound = self.sc.bound
        self.dec = self.dec.to(self.dev)
        for p in list(self.dec.low_decoder.parameters()) + list(self.dec.high_decoder.parameters()):
            p.requires_grad_(False)                       # never in the optimiser (src/Mapper.py:364-371, fix_high: True)
        self.rend = A.Renderer(CFG, None, self.sc)
        self.tsdf_bnds = self.sc.tsdf_bnds.to(self.dev)
        self.bound = self.sc.bound.to(self.dev)
        self.rays, self.total_frames, self.window, self.keyframe_every = rays, total_frames, window, keyframe_every
        gen = torch.Generator().manual_seed(seed)
        sc = self.sc
        self.target_color = torch.rand(sc.H, sc.W, 3, generator=gen).to(self.dev)
        self.c = {k: v.clone() for k, v in sc.c.items()}
        self.keyframes = []
        self.n_iter = 0
        torch.manual_seed(seed)

    # a fixed set of rays from poses BETWEEN the training poses: the held-out depth error of the map
    def heldout_rays(self, frames, n=4000, seed=123, stride=1):
        sc = self.sc
        g = torch.Generator().manual_seed(seed)
        ros, rds, gds = [], [], []
        per = n // frames
        for f in range(frames):
            c2w = circle_pose(sc, (f + 0.5) * stride, self.total_frames)
            depth = sc.depth_image(c2w, zero_band=0.0)
            ro, rd = common.get_rays(sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, c2w, self.dev)
            pick = torch.randint(sc.H * sc.W, (per,), generator=g).to(self.dev)
            ros.append(ro.reshape(-1, 3)[pick]); rds.append(rd.reshape(-1, 3)[pick]); gds.append(depth.reshape(-1)[pick])
        return torch.cat(ros), torch.cat(rds), torch.cat(gds)

    def heldout_error(self, held):
        ro, rd, gd = held
        with torch.no_grad():
            d, _, _, _ = self.rend.render_batch_ray(self.c, self.dec, rd, ro, self.dev, self.sc.tsdf_volume, self.tsdf_bnds,
                                                    'color', gd)
        return float(torch.abs(gd.double() - d).mean())

    def map_frame(self, f, n_iters, lr_factor=1.0, tick=None):
        sc, dev, dec = self.sc, self.dev, self.dec
        H, W, fx, fy, cx, cy = sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy
        c2w = circle_pose(sc, f, self.total_frames)
        depth = sc.depth_image(c2w)
        if f % self.keyframe_every == 0:
            self.keyframes.append((c2w, depth))
        frames = [(c2w, depth)] + self.keyframes[-(self.window - 1):]
        if self.fused:
            return self.map_frame_fused(f, n_iters, lr_factor, c2w, depth, frames)
        grids = {k: v.detach().requires_grad_(True) for k, v in self.c.items()}
        masks = {k: mapping.frustum_mask(c2w, tuple(v.shape[2:]), depth, sc.bound, H, W, fx, fy, cx, cy) for k, v in grids.items()}
        opt_grids = mapping.MaskedGridAdam(grids, masks)
        opt = torch.optim.Adam([{'params': list(dec.color_decoder.parameters()), 'lr': 0},      # fix_high: True
                                {'params': list(dec.mlp.parameters()), 'lr': 0}])
        first = last = None
        for it in range(n_iters):
            stage = stage_of(it, n_iters)
            lr = {k: v * lr_factor for k, v in STAGE_LR[stage].items()}
            for g, key in zip(opt.param_groups, ('dec', 'mlp')):
                g['lr'] = lr[key]
            opt.zero_grad()
            opt_grids.zero_grad()
            ros, rds, gds, gcs = [], [], [], []
            for kc2w, kdepth in frames:
                ro, rd, gd, gc = common.get_samples(0, H, 0, W, self.rays // len(frames), H, W, fx, fy, cx, cy, kc2w,
                                                    kdepth, self.target_color, dev)
                ros.append(ro.float()); rds.append(rd.float()); gds.append(gd.float()); gcs.append(gc.float())
            ro, rd, gd, gc = torch.cat(ros), torch.cat(rds), torch.cat(gds), torch.cat(gcs)
            ro, rd, gd, gc = common.filter_rays_in_bound(ro, rd, gd, gc, self.bound)     # src/Mapper.py:439-449
            d, u, col, w = self.rend.render_batch_ray(grids, dec, rd, ro, dev, sc.tsdf_volume, self.tsdf_bnds, stage, gd)
            loss, m = mapper_loss(stage, it, n_iters, f, gd, gc, d, col, w)
            loss.backward()
            opt.step()
            opt_grids.step({'grid_low': lr['low'], 'grid_high': lr['high'], 'grid_color': lr['color']})
            self.n_iter += 1
            if it == 0 or it == n_iters - 1:
                per_ray = float(torch.abs(gd[m] - d[m].detach()).sum()) / max(1, int(m.sum()))
                first = per_ray if it == 0 else first
                last = per_ray
        self.c = {k: v.detach() for k, v in grids.items()}
        if not math.isfinite(last):
            raise RuntimeError(f'non-finite loss at frame {f}')
        return first, last


def _map_frame_fused(self, f, n_iters, lr_factor, c2w, depth, frames):
    """The same frame through mapping.MapperIteration: pre-filter as a keep mask, loss, backward and Adam on the device, one
    graph replay per iteration, no host read-back inside the loop."""
    sc, dev = self.sc, self.dev
    H, W, fx, fy, cx, cy = sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy
    masks = {k: mapping.frustum_mask(c2w, tuple(v.shape[2:]), depth, sc.bound, H, W, fx, fy, cx, cy) for k, v in self.c.items()}
    lrs = {st: {'low': v['low'] * lr_factor, 'high': v['high'] * lr_factor, 'color': v['color'] * lr_factor,
                'decoders': v['dec'] * lr_factor, 'mlp': v['mlp'] * lr_factor} for st, v in STAGE_LR.items()}
    key = float(lr_factor)
    if self.iteration is None:
        self.iteration = {}
        self.c = {k: v.detach().clone().contiguous() for k, v in self.c.items()}
    it = self.iteration.get(key)
    if it is None:
        it = self.iteration[key] = mapping.MapperIteration(self.rend, self.dec, self.c, masks, sc.tsdf_volume, self.tsdf_bnds, lrs,
                                                         w_color_loss=W_COLOR_LOSS)
    it.new_frame(masks)
    low_end = int(n_iters * LOW_ITER_RATIO)
    first = last = None
    for i in range(n_iters):
        stage = stage_of(i, n_iters)
        # the window's rays (src/Mapper.py:421-436: get_samples per keyframe + four torch.cat) straight into the iteration's input buffers
        n_each = self.rays // len(frames)
        batch = common.get_samples_multi(0, H, 0, W, n_each, H, W, fx, fy, cx, cy, [(kc2w, kdepth, self.target_color) for kc2w, kdepth in frames],
                                         dev, out=it.input_buffers(n_each * len(frames)))
        loss = it.step(*batch, stage, warmup=(low_end < i <= low_end + 5 and f <= 1))
        self.n_iter += 1
        if i == 0 or i == n_iters - 1:
            n_rays = n_each * len(frames)
            v = float(loss) / n_rays                       # total loss per ray (the loop's only read-backs: first and last iteration)
            first = v if i == 0 else first
            last = v
    if not math.isfinite(last):
        raise RuntimeError(f'non-finite loss at frame {f}')
    return first, last


MappingRun.map_frame_fused = _map_frame_fused


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=200)
    ap.add_argument('--iters', type=int, default=60)              # configs/df_prior.yaml:63
    ap.add_argument('--iters-first', type=int, default=300)       # the reference uses 1500 (:64)
    ap.add_argument('--rays', type=int, default=5000)
    ap.add_argument('--scene', default='office0')
    ap.add_argument('--fused', action='store_true', help='mapping.MapperIteration (device-side iteration, graph replay)')
    ap.add_argument('--every-frame', type=int, default=1, help='map every n-th frame of the sequence (configs/df_prior.yaml:44: 5)')
    args = ap.parse_args()
    run = MappingRun(args.scene, args.rays, args.frames, fused=args.fused)
    calls = list(range(0, args.frames, args.every_frame))
    held = run.heldout_rays(min(len(calls), 40), stride=args.every_frame)
    e0 = run.heldout_error(held)
    hist = []
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for f in calls:
        hist.append(run.map_frame(f, args.iters_first if f == 0 else args.iters, LR_FIRST_FACTOR if f == 0 else 1.0))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t_start
    e1 = run.heldout_error(held)
    q = max(1, len(hist) // 4)
    mean = lambda xs: sum(xs) / len(xs)
    print(json.dumps({'config': f'{args.scene} mapping loop', 'frames': args.frames, 'mapping_calls': len(calls), 'iters_per_frame': args.iters,
                      'iters_first': args.iters_first, 'rays_per_iter': args.rays, 'fused': args.fused, 'iterations': run.n_iter, 'seconds': dt,
                      'ms_per_iteration': dt / run.n_iter * 1e3,
                      'heldout_depth_l1_before': e0, 'heldout_depth_l1_after': e1,
                      'depth_loss_per_ray_first_quarter': mean([h[1] for h in hist[:q]]),
                      'depth_loss_per_ray_last_quarter': mean([h[1] for h in hist[-q:]]),
                      'all_finite': True, 'grid_absmax': {k: float(v.abs().max()) for k, v in run.c.items()}}))


if __name__ == '__main__':
    main()
