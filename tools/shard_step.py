"""One rank's share of the ray-sharded headline frame, run the way `bench.py --gpus k` runs it (caches cleared per step, steps back
to back, ONE synchronisation at the end) -- for a kernel trace of a 1/k shard on ONE GPU.

  python tools/shard_step.py                         pipelined ms per step of every shard of k = 1, 2, 4, 8
  rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 tools/shard_step.py --trace 8 3
                                                     20 steady-state steps of shard 3 of 8 only (then tools/trace_timeline.py
                                                     <kernel_trace.csv> k_get_rays for one step's timeline)
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B                                                    # noqa: E402
import attentive_dfprior_amd as A                                    # noqa: E402
from attentive_dfprior_amd import synthetic, dist as adist           # noqa: E402


def main():
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    scene, sd, dec = B.build_scene(A, synthetic, 'room0', dev)
    rend = A.Renderer(B.CFG64, None, scene)
    tsdf_bnds = scene.tsdf_bnds.to(dev)
    c2w = scene.default_c2w(yaw=0.3, pitch=-0.1)
    gt_depth = scene.depth_image(c2w)
    n = scene.H * scene.W

    def step(lo, hi):
        rend._engine._grid_cache.clear()
        dec._packed.clear()
        return rend.render_img_shard(scene.c, dec, c2w, dev, scene.tsdf_volume, tsdf_bnds, 'color', gt_depth, lo, hi)

    def pipelined(lo, hi, steps=20, warm=3):
        for _ in range(warm):
            step(lo, hi)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            step(lo, hi)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / steps * 1e3

    if len(sys.argv) > 1 and sys.argv[1] == '--trace':
        k, r = int(sys.argv[2]), int(sys.argv[3])
        lo, hi = adist.shard_range(n, r, k)
        print(f'shard {r} of {k}: rays [{lo}, {hi}): {pipelined(lo, hi):.4f} ms per step (pipelined, 20 steps)')
        return
    for rep in range(2):
        for k in (1, 2, 4, 8):
            ts = [pipelined(*adist.shard_range(n, r, k)) for r in range(k)]
            print(f'k = {k}: pipelined ms per step, slowest {max(ts):.4f} fastest {min(ts):.4f}  (linear share of k = 1 would be shown below)')
    t1 = pipelined(0, n)
    print(f'k = 1 again: {t1:.4f} ms;  t1 / 8 = {t1 / 8:.4f} ms')


if __name__ == '__main__':
    main()
