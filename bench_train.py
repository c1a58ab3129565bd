"""Secondary benchmark (not the driver's contract line): one Mapper iteration =
render_batch_ray forward + Mapper loss + backward (src/Mapper.py:451-473) on the room0-sized
synthetic scene.  Prints one JSON line per configuration.

  python bench_train.py [--rays 1000 5000] [--iters 20]
"""
import argparse
import json
import time

import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import synthetic


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rays', type=int, nargs='+', default=[1000, 5000])
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--scene', default='room0')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    scene = synthetic.Scene(args.scene, device=dev, grid_std_scale=20.0)
    scene.c['grid_high'] = scene.c['grid_high'] * 100
    dec = A.DF()
    dec.load_state_dict(synthetic.seeded_state_dict(0))
    dec.bound = scene.bound
    dec = dec.to(dev)
    for p in list(dec.low_decoder.parameters()) + list(dec.high_decoder.parameters()):
        p.requires_grad_(False)                       # low never optimised, fix_high: True
    tsdf_bnds = scene.tsdf_bnds.to(dev)
    c2w = scene.default_c2w()
    gt_depth = scene.depth_image(c2w)
    from attentive_dfprior_amd.common import get_rays
    ro_all, rd_all = get_rays(scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy, c2w, dev)
    ro_all, rd_all, gd_all = ro_all.reshape(-1, 3), rd_all.reshape(-1, 3), gt_depth.reshape(-1)
    for n_rays in args.rays:
        for ns, nf in ((32, 16), (48, 16)):
            cfg = {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': ns, 'N_surface': nf, 'N_importance': 0},
                   'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}
            rend = A.Renderer(cfg, None, scene)
            c = {k: v.clone().requires_grad_(True) for k, v in scene.c.items()}
            params = list(dec.color_decoder.parameters()) + list(dec.mlp.parameters())
            opt = torch.optim.Adam([{'params': params, 'lr': 0.005}, {'params': list(c.values()), 'lr': 0.005}])
            g = torch.Generator(device='cpu').manual_seed(0)
            pick = torch.randint(ro_all.shape[0], (n_rays,), generator=g).to(dev)
            ro, rd, gd = ro_all[pick], rd_all[pick], gd_all[pick]
            gc = torch.rand(n_rays, 3, device=dev)

            def it(step=True):
                opt.zero_grad()
                d, u, col, w = rend.render_batch_ray(c, dec, rd, ro, dev, scene.tsdf_volume, tsdf_bnds, 'color', gt_depth=gd)
                m = gd > 0
                loss = torch.abs(gd[m] - d[m]).sum() + 0.2 * torch.abs(gc - col).sum()
                loss.backward()
                if step:
                    opt.step()
                return loss
            for _ in range(3):
                it()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                loss = it()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.iters
            t0 = time.perf_counter()
            with torch.no_grad():
                for _ in range(args.iters):
                    rend.render_batch_ray(c, dec, rd, ro, dev, scene.tsdf_volume, tsdf_bnds, 'color', gt_depth=gd)
            torch.cuda.synchronize()
            df = (time.perf_counter() - t0) / args.iters
            # the same iteration as one fused, graph-replayed call (mapping.MapperIteration): pre-filter mask, render, loss,
            # backward and Adam with no host read-back
            from attentive_dfprior_amd import mapping
            import copy
            dec_f = copy.deepcopy(dec)
            cf = {k: v.detach().clone() for k, v in scene.c.items()}
            lr = {'color': dict(low=0.005, high=0.005, color=0.005, decoders=0.005, mlp=0.005)}
            res = {}
            for mode, use_graph in (('fused_eager', False), ('fused_graph', True)):
                itf = mapping.MapperIteration(A.Renderer(cfg, None, scene), dec_f, cf, None, scene.tsdf_volume, tsdf_bnds, lr, use_graph=use_graph)
                for _ in range(3):
                    itf.step(ro, rd, gd, gc, 'color')
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.iters):
                    itf.step(ro, rd, gd, gc, 'color')
                torch.cuda.synchronize()
                res[mode] = (time.perf_counter() - t0) / args.iters * 1e3
            # the floor the CALLER's own torch code sets (tools/host_breakdown.py: render_batch_ray replaced by an allocation-only stub wired
            # into autograd like the real function): zero_grad, the loss ops with their boolean-index syncs, one AccumulateGrad per
            # parameter, torch.optim.Adam over the same 36 tensors -- what an unchanged Mapper.py pays whatever the renderer costs
            import os, sys
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))
            from host_breakdown import stub_render_batch_ray
            real = rend.render_batch_ray
            rend.render_batch_ray = stub_render_batch_ray(rend, dec)
            try:
                for _ in range(5):
                    it()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.iters):
                    it()
                torch.cuda.synchronize()
                floor = (time.perf_counter() - t0) / args.iters
            finally:
                rend.render_batch_ray = real
            print(json.dumps({'metric': 'mapper iteration (render fwd + loss + bwd + Adam), stage color', 'rays': n_rays,
                              'ms_per_iter_torch_floor': floor * 1e3, 'ms_per_iter_above_floor': (dt - floor) * 1e3,
                              'samples_per_ray': ns + nf, 'ms_per_iter': dt * 1e3, 'rays_per_s_fwd_bwd': n_rays / dt,
                              'ms_forward_only': df * 1e3, 'loss': float(loss),
                              'ms_per_iter_fused_eager': res['fused_eager'], 'ms_per_iter_fused_graph': res['fused_graph']}))


if __name__ == '__main__':
    main()
