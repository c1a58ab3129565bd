"""North-star target check: >= 10x the reference's single-GPU PyTorch path on the same MI355X.

The denominator is the oracle (== the reference's own PyTorch ops, pinned bit-for-bit on CPU) run
with its tensors on the GPU through PyTorch-ROCm, chunked like Renderer.eval_points chunks
(points_batch_size 500 000, Renderer.py:38).  Same rays, same grids, same weights; the product path is
timed on the whole reference ray batch (100 000 rays x 64 samples).  The GPU-torch result is also a
second, full-size parity reference (torch's GPU sin/grid_sample are not bit-identical to the CPU ops,
hence its own tolerance).
"""
import json
import time

import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import synthetic
from attentive_dfprior_amd.common import get_rays
from oracle import adfp_oracle as O

pytestmark = pytest.mark.gpu


def _torch_gpu_render(sd, c, rd, ro, tsdf, tsdf_bnds, bound, gd, NS, NF, chunk_points=500000):
    """Renderer.render_batch_ray with eval_points' 500k-point chunk loop, all on the GPU in torch."""
    z = O.sample_z(ro, rd, gd, bound, NS, NF, False, 0.0, None, None)
    N, S = z.shape
    pts = (ro[..., None, :] + rd[..., None, :] * z[..., :, None]).reshape(-1, 3)
    raws, ws = [], []
    for i in range(0, pts.shape[0], chunk_points):
        r, w = O.eval_points(sd, pts[i:i + chunk_points], c, tsdf, tsdf_bnds, bound, 'color')
        raws.append(r)
        ws.append(w)
    raw = torch.cat(raws).reshape(N, S, 4)
    depth, var, color, _ = O.raw2outputs(raw, z)
    return depth, var, color, torch.cat(ws).reshape(N, S, 1)


def test_ten_times_single_gpu_pytorch():
    dev = torch.device('cuda:0')
    NS, NF, N = 48, 16, 100000
    scene = synthetic.Scene('room0', device=dev, grid_std_scale=20.0)
    scene.c['grid_high'] = scene.c['grid_high'] * 100
    sd = synthetic.seeded_state_dict(0)
    dec = A.DF()
    dec.load_state_dict(sd)
    dec.bound = scene.bound
    dec = dec.to(dev)
    cfg = {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': NS, 'N_surface': NF, 'N_importance': 0},
           'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}
    rend = A.Renderer(cfg, None, scene)
    tsdf_bnds = scene.tsdf_bnds.to(dev)
    c2w = scene.default_c2w()
    gd_img = scene.depth_image(c2w)
    ro, rd = get_rays(scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy, c2w, dev)
    pick = torch.arange(0, scene.H * scene.W, 3, device=dev)[:N]
    ro, rd, gd = ro.reshape(-1, 3)[pick].contiguous(), rd.reshape(-1, 3)[pick].contiguous(), gd_img.reshape(-1)[pick].contiguous()

    def product():
        with torch.no_grad():
            return rend.render_batch_ray(scene.c, dec, rd, ro, dev, scene.tsdf_volume, tsdf_bnds, 'color', gt_depth=gd)

    sd_g = {k: v.to(dev) for k, v in sd.items()}
    bound_g = scene.bound.to(dev)

    def torch_gpu():
        with torch.no_grad():
            return _torch_gpu_render(sd_g, scene.c, rd, ro, scene.tsdf_volume, tsdf_bnds, bound_g, gd, NS, NF)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, out

    t_p, (d, u, c, w) = timed(product, 10)
    t_t, (od, ou, oc, ow) = timed(torch_gpu, 3)
    ratio = t_t / t_p
    rel_d = float(((d - od).abs() / od.abs().clamp_min(1e-3)).max())
    rel_c = float(((c - oc).abs().max()) / oc.abs().max())
    print(json.dumps({'rays': N, 'samples_per_ray': NS + NF, 'product_ms': t_p * 1e3, 'torch_gpu_ms': t_t * 1e3,
                      'product_rays_per_s': N / t_p, 'torch_gpu_rays_per_s': N / t_t, 'speedup': ratio,
                      'max_rel_depth': rel_d, 'max_rel_color': rel_c}))
    assert ratio >= 10.0, f'only {ratio:.1f}x the PyTorch-ROCm path'
    # full-size parity against torch on the GPU (north star: <= 1e-4 relative)
    assert rel_d <= 1e-4 and rel_c <= 1e-4
