"""Config 5 smoke (BASELINE.json configs[4], one GPU's share): 16 m cube, 1024^3 TSDF (4.3 GB),
128 samples/ray (96 + 32), rays from several poses.  Prints rays/s; the parity check of the same
configuration is tests/test_gpu_config5.py.  Not part of the driver contract."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import attentive_dfprior_amd as A                      # noqa: E402
from attentive_dfprior_amd import synthetic            # noqa: E402
from attentive_dfprior_amd.common import get_rays      # noqa: E402


def main(n_rays=131072, order='pixel'):
    dev = torch.device('cuda:0')
    sc = synthetic.Scene('cube16', device=dev, grid_std_scale=20.0, voxel=16.0 / 1024, inset=2.0)
    sc.c['grid_high'] = sc.c['grid_high'] * 100
    assert sc.tsdf_volume.numel() >= 1000 ** 3
    sd = synthetic.seeded_state_dict(0)
    dec = A.DF(); dec.load_state_dict(sd); dec.bound = sc.bound; dec = dec.to(dev)
    cfg = {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': 96, 'N_surface': 32, 'N_importance': 0},
           'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}
    rend = A.Renderer(cfg, None, sc)
    tsdf_bnds = sc.tsdf_bnds.to(dev)
    ros, rds, gds = [], [], []
    for k in range(8):
        c2w = sc.default_c2w(offset=(0.5 * k - 2, 0.3 * k - 1, 0.2 * k), yaw=0.7 * k, pitch=-0.2 + 0.05 * k)
        gd = sc.depth_image(c2w)
        ro, rd = get_rays(sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, c2w, dev)
        per = n_rays // 8
        pick = torch.randperm(sc.H * sc.W, device=dev)[:per] if order.startswith('random') else torch.arange((sc.H * sc.W - per) // 2, (sc.H * sc.W - per) // 2 + per, device=dev)
        ros.append(ro.reshape(-1, 3)[pick]); rds.append(rd.reshape(-1, 3)[pick]); gds.append(gd.reshape(-1)[pick])
    ro, rd, gd = torch.cat(ros), torch.cat(rds), torch.cat(gds)
    if order == 'random-sorted':
        # experiment: the random draw, then ordered by (coarse origin cell, fine cell of the surface point) -- what a ray sort in front of
        # the TSDF stage could give at best (the sort itself is NOT timed here)
        lo, hi = sc.tsdf_bnds[:, 0].to(dev).float(), sc.tsdf_bnds[:, 1].to(dev).float()

        def cell(p, bits):
            q = ((p - lo) / (hi - lo) * (1 << bits)).long().clamp(0, (1 << bits) - 1)
            key = torch.zeros(p.shape[0], dtype=torch.long, device=dev)
            for b in range(bits):
                for ax in range(3):
                    key |= ((q[:, ax] >> b) & 1) << (3 * b + ax)
            return key
        surf = ro + rd * torch.where(gd > 0, gd, torch.ones_like(gd))[:, None]
        key = (cell(ro, 3) << 24) | cell(surf, 8)
        perm = torch.argsort(key)
        if os.environ.get('SORT_LIB'):                     # the library's own keys + radix sort (Renderer._coherent_order)
            perm = rend._coherent_order(ro.contiguous(), rd.contiguous(), gd.contiguous(), sc.tsdf_volume, tsdf_bnds, wait=True)
        ro, rd, gd = ro[perm].contiguous(), rd[perm].contiguous(), gd[perm].contiguous()
    if order != 'random':
        rend.sort_rays_min = 0                             # 'random' alone goes through the renderer's automatic path
    with torch.no_grad():
        for _ in range(3):          # the second call of a like batch acts on the first one's order verdict; the third has its buffers
            out = rend.render_batch_ray(sc.c, dec, rd, ro, dev, sc.tsdf_volume, tsdf_bnds, 'color', gt_depth=gd)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = rend.render_batch_ray(sc.c, dec, rd, ro, dev, sc.tsdf_volume, tsdf_bnds, 'color', gt_depth=gd)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
    d, u, c, w = out
    assert torch.isfinite(d).all() and torch.isfinite(c).all()
    res = {'config': '1024^3 TSDF, 128 samples/ray', 'ray_order': order, 'rays': ro.shape[0], 'ms': dt * 1e3, 'rays_per_s': ro.shape[0] / dt,
           'tsdf_GB': sc.tsdf_volume.numel() * 4 / 1e9, 'in_band_fraction': float((w != 1).float().mean())}
    print(json.dumps(res))


if __name__ == '__main__':
    main(order=sys.argv[1] if len(sys.argv) > 1 else 'pixel')
