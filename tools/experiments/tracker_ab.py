"""A/B of the fused Tracker iteration (graph replay) between the in-tree library and the one ADFP_LIB_PATH names."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import attentive_dfprior_amd as A
from attentive_dfprior_amd import synthetic
from attentive_dfprior_amd.tracking import TrackerIteration
import bench
import bench_extra as BX
dev = torch.device('cuda:0')
scene, sd, dec = bench.build_scene(A, synthetic, 'room0', dev)
for p in dec.parameters():
    p.requires_grad_(False)
rend = A.Renderer(BX._cfg(48, 16), None, scene)
c2w = scene.default_c2w()
depth = scene.depth_image(c2w)
color = torch.rand((scene.H, scene.W, 3), generator=torch.Generator().manual_seed(0)).to(dev)
tb = scene.tsdf_bnds.to(dev)
for n in (200, 1000):
    it = TrackerIteration(rend, dec, scene.c, scene.tsdf_volume, tb, scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy, 20, 20)
    cam = BX._tensor_from_c2w(c2w).to(dev); cam[4:] += 0.01
    it.new_frame(cam, depth, color)
    for _ in range(20):
        it.step(n)
    torch.cuda.synchronize()
    for rep in range(3):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(300):
            it.step(n)
        e1.record(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        print(f'tracker {n} rays: {e0.elapsed_time(e1) / 300:.4f} ms per iteration by events, {(t1 - t0) / 300 * 1e3:.4f} ms wall ({os.environ.get("ADFP_LIB_PATH", "in-tree")})')
