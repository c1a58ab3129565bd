"""Host-side cost of one training iteration: with 64 rays the GPU work is negligible, so the wall time per
section is what Python / ctypes / autograd / allocator spend enqueueing it.  Not part of the driver contract."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import attentive_dfprior_amd as A                      # noqa: E402
from attentive_dfprior_amd import synthetic            # noqa: E402


def main(n_rays=64, iters=200):
    dev = torch.device('cuda:0')
    sc = synthetic.Scene('room0', device=dev)
    dec = A.DF()
    dec.load_state_dict(synthetic.seeded_state_dict(0))
    dec.bound = sc.bound
    dec = dec.to(dev)
    for p in list(dec.low_decoder.parameters()) + list(dec.high_decoder.parameters()):
        p.requires_grad_(False)
    cfg = {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': 48, 'N_surface': 16, 'N_importance': 0},
           'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}
    rend = A.Renderer(cfg, None, sc)
    tsdf_bnds = sc.tsdf_bnds.to(dev)
    ro, rd, gd, gc = (t.to(dev) for t in synthetic.make_ray_batch(sc, n_rays, seed=1))
    c = {k: v.clone().requires_grad_(True) for k, v in sc.c.items()}
    opt = torch.optim.Adam([{'params': list(dec.color_decoder.parameters()) + list(dec.mlp.parameters()), 'lr': 1e-3},
                            {'params': list(c.values()), 'lr': 1e-3}])
    sect = {}

    def run(sync):
        for it in range(iters):
            t0 = time.perf_counter()
            opt.zero_grad()
            d, u, col, w = rend.render_batch_ray(c, dec, rd, ro, dev, sc.tsdf_volume, tsdf_bnds, 'color', gt_depth=gd)
            if sync: torch.cuda.synchronize()
            t1 = time.perf_counter()
            m = gd > 0
            loss = torch.abs(gd[m] - d[m]).sum() + 0.2 * torch.abs(gc - col).sum()
            if sync: torch.cuda.synchronize()
            t2 = time.perf_counter()
            loss.backward()
            if sync: torch.cuda.synchronize()
            t3 = time.perf_counter()
            opt.step()
            if sync: torch.cuda.synchronize()
            t4 = time.perf_counter()
            if it >= 20:
                for k, v in (('forward', t1 - t0), ('loss', t2 - t1), ('backward', t3 - t2), ('adam', t4 - t3)):
                    sect[k] = sect.get(k, 0.0) + v
        torch.cuda.synchronize()
    for sync in (False, True):
        sect.clear()
        run(sync)
        print(json.dumps({'rays': n_rays, 'synchronised_sections': sync,
                          'ms': {k: v / (iters - 20) * 1e3 for k, v in sect.items()}}))


if __name__ == '__main__':
    main()
