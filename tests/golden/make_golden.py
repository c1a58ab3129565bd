"""
Generates the golden fixtures under tests/golden/ by running the REFERENCE's own code
(imported read-only from /root/reference through oracle/ref_import.py) on seeded inputs.
Build container only; the fixtures it writes are data (inputs + expected outputs) and are
what travels to the GPU box.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Files
  mini_inputs.npz    scene (bound, tsdf volume in its physical [X,Y,Z] order + bounds, the three
                     feature grids), decoder state dict, ray batch (rays_o, rays_d, gt_depth with
                     zero-depth rays, gt_color), camera pose + intrinsics, explicit query points
  mini_<stage>.npz   reference outputs for stage in low/high/color:
                     render_batch_ray -> depth, uncertainty, color, weight (+ dtypes),
                     intermediates z_vals, raw (captured at the compositing call),
                     render without sensor depth (nd_*), eval_points on explicit points,
                     Mapper-loss gradients w.r.t. the grids and every decoder parameter
                     (plain loss and the warm-up variant with the |w-1| term)
  mini_tracker.npz   Tracker loss (src/Tracker.py:116-129) and its gradients w.r.t. rays_o / rays_d
  mini_rays.npz      get_rays / get_rays_from_uv vectors, TSDF point samples, render_img tile
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import adfp_oracle as O          # noqa: E402  (only for the seeded state dict helper)
from oracle import ref_import                # noqa: E402
from attentive_dfprior_amd import synthetic  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
N_SAMPLES, N_SURFACE, N_RAYS = 32, 16, 160


def main():
    scene = synthetic.mini_scene()
    sd = O.random_state_dict(seed=3)
    rays_o, rays_d, depth, color = synthetic.make_ray_batch(scene, N_RAYS, seed=5)
    g = torch.Generator().manual_seed(9)
    # explicit points: inside, on the band, outside bound, outside the TSDF volume
    lo, hi = scene.bound[:, 0], scene.bound[:, 1]
    qp = lo + (hi - lo) * (torch.rand(400, 3, generator=g, dtype=torch.float64) * 1.2 - 0.1)
    c2w = scene.default_c2w(offset=(0.05, -0.03, 0.02), yaw=0.7, pitch=0.15)
    depth_img = scene.depth_image(c2w, zero_band=0.1)

    tsdf_phys = scene.tsdf_volume.permute(0, 1, 4, 3, 2).contiguous()[0, 0]   # [X,Y,Z]
    inputs = {'bound': scene.bound.numpy(), 'tsdf_bnds': scene.tsdf_bnds.numpy(), 'tsdf_xyz': tsdf_phys.numpy(),
              'rays_o': rays_o.numpy(), 'rays_d': rays_d.numpy(), 'gt_depth': depth.numpy(), 'gt_color': color.numpy(),
              'query_points': qp.numpy(), 'c2w': c2w.numpy(), 'depth_img': depth_img.numpy(),
              'intrinsics': np.array([scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy], dtype=np.float64),
              'n_samples': np.array(N_SAMPLES), 'n_surface': np.array(N_SURFACE)}
    for k, v in scene.c.items():
        inputs[k] = v.numpy()
    for k, v in sd.items():
        inputs['sd.' + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, 'mini_inputs.npz'), **inputs)

    df, rend, rcommon = ref_import.make_reference_objects(scene, sd, N_SAMPLES, N_SURFACE)
    _, _, rrend = ref_import.load()

    captured = {}
    orig_r2o = rrend.raw2outputs_nerf_color

    def capture(raw, z_vals, rays_d_, occupancy=False, device='cpu'):
        captured['raw'] = raw.detach().clone()
        captured['z_vals'] = z_vals.detach().clone()
        return orig_r2o(raw, z_vals, rays_d_, occupancy=occupancy, device=device)

    for stage in ('low', 'high', 'color'):
        out = {}
        rrend.raw2outputs_nerf_color = capture
        with torch.no_grad():
            d, u, col, w = rend.render_batch_ray(scene.c, df, rays_d, rays_o, 'cpu', scene.tsdf_volume,
                                                 scene.tsdf_bnds, stage, gt_depth=depth)
            out.update(depth=d.numpy(), uncertainty=u.numpy(), color=col.numpy(), weight=w.numpy(),
                       z_vals=captured['z_vals'].numpy(), raw=captured['raw'].numpy())
            d, u, col, w = rend.render_batch_ray(scene.c, df, rays_d, rays_o, 'cpu', scene.tsdf_volume,
                                                 scene.tsdf_bnds, stage, gt_depth=None)
            out.update(nd_depth=d.numpy(), nd_uncertainty=u.numpy(), nd_color=col.numpy(), nd_weight=w.numpy(),
                       nd_z_vals=captured['z_vals'].numpy())
            raw_q, w_q = rend.eval_points(qp, df, scene.tsdf_volume, scene.tsdf_bnds, scene.c, stage, 'cpu')
            out.update(q_raw=raw_q.numpy(), q_w=w_q.numpy())
            raw_d, w_d = df(qp.unsqueeze(0), c_grid=scene.c, tsdf_volume=scene.tsdf_volume,
                            tsdf_bnds=scene.tsdf_bnds, stage=stage)
            out.update(df_raw=raw_d.numpy(), df_w=w_d.numpy())
        rrend.raw2outputs_nerf_color = orig_r2o
        # Mapper loss gradients (src/Mapper.py:457-473)
        for tag, warm in (('g', False), ('gw', True)):
            c_req = {k: v.clone().requires_grad_(True) for k, v in scene.c.items()}
            for p in df.parameters():
                p.requires_grad_(True)
                p.grad = None
            d, u, col, w = rend.render_batch_ray(c_req, df, rays_d, rays_o, 'cpu', scene.tsdf_volume,
                                                 scene.tsdf_bnds, stage, gt_depth=depth)
            m = depth > 0
            loss = torch.abs(depth[m] - d[m]).sum()
            if warm:
                loss = loss + torch.abs(w - torch.ones(w.shape)).sum()
            if stage == 'color':
                loss = loss + 0.2 * torch.abs(color - col).sum()
            loss.backward()
            out[tag + '.loss'] = np.array(loss.item())
            for k, v in c_req.items():
                out[f'{tag}.{k}'] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy()
            for name, p in df.named_parameters():
                out[f'{tag}.sd.{name}'] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy()
            for p in df.parameters():
                p.requires_grad_(False)
        np.savez_compressed(os.path.join(OUT, f'mini_{stage}.npz'), **out)
        print(stage, 'depth', out['depth'][:4], 'loss', out['g.loss'])

    # Tracker: gradients w.r.t. the rays from the REFERENCE (src/Tracker.py:112-133)
    tr = {}
    ro_g = rays_o.clone().requires_grad_(True)
    rd_g = rays_d.clone().requires_grad_(True)
    d, u, col, w = rend.render_batch_ray(scene.c, df, rd_g, ro_g, 'cpu', scene.tsdf_volume, scene.tsdf_bnds, 'color',
                                         gt_depth=depth)
    u = u.detach()
    tmp = torch.abs(depth - d) / torch.sqrt(u + 1e-10)
    m = (tmp < 10 * tmp.median()) & (depth > 0)
    loss = (torch.abs(depth - d) / torch.sqrt(u + 1e-10))[m].sum() + 0.5 * torch.abs(color - col)[m].sum()
    loss.backward()
    tr['loss'] = np.array(loss.item())
    tr['g_rays_o'], tr['g_rays_d'] = ro_g.grad.numpy(), rd_g.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'mini_tracker.npz'), **tr)
    print('tracker loss', tr['loss'], 'max |g_o|', np.abs(tr['g_rays_o']).max(), 'max |g_d|', np.abs(tr['g_rays_d']).max())

    # rays + tsdf samples + full image tile
    rr = {}
    ro, rd = rcommon.get_rays(scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy, c2w, 'cpu')
    rr['get_rays_o'], rr['get_rays_d'] = ro.numpy(), rd.numpy()
    ii = torch.tensor([0., 5., 63., 31.])
    jj = torch.tensor([0., 47., 2., 23.])
    ro2, rd2 = rcommon.get_rays_from_uv(ii, jj, c2w, scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy, 'cpu')
    rr['uv_i'], rr['uv_j'], rr['uv_rays_o'], rr['uv_rays_d'] = ii.numpy(), jj.numpy(), ro2.numpy(), rd2.numpy()
    with torch.no_grad():
        rr['tsdf_q'] = rend.eval_points_tsdf(qp, scene.tsdf_volume, 'cpu').numpy()
        rend.ray_batch_size = 1000          # exercise the per-batch far clamp of render_img
        di, ui, ci = rend.render_img(scene.c, df, c2w, 'cpu', scene.tsdf_volume, scene.tsdf_bnds, 'color',
                                     gt_depth=depth_img)
    rr['img_depth'], rr['img_uncertainty'], rr['img_color'] = di.numpy(), ui.numpy(), ci.numpy()
    rr['img_ray_batch_size'] = np.array(1000)
    np.savez_compressed(os.path.join(OUT, 'mini_rays.npz'), **rr)
    print('done')


if __name__ == '__main__':
    main()
