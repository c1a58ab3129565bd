"""A small forward render (the Tracker's batch size) captured into a HIP graph with torch.cuda.graph and
replayed with new rays: adfp_render_forward only enqueues kernels on the caller's stream (no allocation, no
host read-back, zero-fills are kernels), so the whole call is capturable.  Prints eager vs replay time.
Not part of the driver contract."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import attentive_dfprior_amd as A                      # noqa: E402
from attentive_dfprior_amd import synthetic            # noqa: E402


def main(n_rays=200):
    dev = torch.device('cuda:0')
    sc = synthetic.Scene('room0', device=dev)
    dec = A.DF()
    dec.load_state_dict(synthetic.seeded_state_dict(0))
    dec.bound = sc.bound
    dec = dec.to(dev)
    cfg = {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': 48, 'N_surface': 16, 'N_importance': 0},
           'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}
    rend = A.Renderer(cfg, None, sc)
    tsdf_bnds = sc.tsdf_bnds.to(dev)
    ro, rd, gd, _ = (t.to(dev) for t in synthetic.make_ray_batch(sc, n_rays, seed=1))
    ro2, rd2, gd2, _ = (t.to(dev) for t in synthetic.make_ray_batch(sc, n_rays, seed=2))

    def render(o, d, z):
        return rend.render_batch_ray(sc.c, dec, d, o, dev, sc.tsdf_volume, tsdf_bnds, 'color', gt_depth=z)

    with torch.no_grad():
        ref2 = render(ro2, rd2, gd2)
        s_ro, s_rd, s_gd = ro.clone(), rd.clone(), gd.clone()
        for _ in range(2):
            render(s_ro, s_rd, s_gd)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = render(s_ro, s_rd, s_gd)
        s_ro.copy_(ro2)
        s_rd.copy_(rd2)
        s_gd.copy_(gd2)
        g.replay()
        torch.cuda.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(out, ref2))

        def timeit(fn, n=500):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        print(json.dumps({'rays': n_rays, 'samples_per_ray': 64, 'replay_equals_eager_bitwise': same,
                          'eager_ms': timeit(lambda: render(ro, rd, gd)), 'graph_replay_ms': timeit(g.replay)}))


if __name__ == '__main__':
    main()
