"""GPU: BASELINE.json configs[4] at one GPU's share -- 16 m cube, 1024^3 TSDF (4.3 GB, beyond every cache),
128 samples/ray (96 + 32), rays of 8 poses -- checked on a ray subset against the oracle (which needs the
whole volume on the host, hence the subset)."""
import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import synthetic
from attentive_dfprior_amd.common import get_rays
from oracle import adfp_oracle as O

pytestmark = pytest.mark.gpu


def test_1024_cubed_tsdf_128_samples_vs_oracle():
    dev = torch.device('cuda:0')
    n_rays, check = 131072, 300
    sc = synthetic.Scene('cube16', device=dev, grid_std_scale=20.0, voxel=16.0 / 1024, inset=2.0)
    sc.c['grid_high'] = sc.c['grid_high'] * 100
    assert sc.tsdf_volume.numel() >= 1000 ** 3
    sd = synthetic.seeded_state_dict(0)
    dec = A.DF()
    dec.load_state_dict(sd)
    dec.bound = sc.bound
    dec = dec.to(dev)
    cfg = {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': 96, 'N_surface': 32, 'N_importance': 0},
           'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}
    rend = A.Renderer(cfg, None, sc)
    tsdf_bnds = sc.tsdf_bnds.to(dev)
    ros, rds, gds = [], [], []
    g = torch.Generator(device='cpu').manual_seed(0)
    for k in range(8):
        c2w = sc.default_c2w(offset=(0.5 * k - 2, 0.3 * k - 1, 0.2 * k), yaw=0.7 * k, pitch=-0.2 + 0.05 * k)
        gd = sc.depth_image(c2w)
        ro, rd = get_rays(sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, c2w, dev)
        pick = torch.randperm(sc.H * sc.W, generator=g)[:n_rays // 8].to(dev)
        ros.append(ro.reshape(-1, 3)[pick])
        rds.append(rd.reshape(-1, 3)[pick])
        gds.append(gd.reshape(-1)[pick])
    ro, rd, gd = torch.cat(ros), torch.cat(rds), torch.cat(gds)
    with torch.no_grad():
        for _ in range(2):                                         # the second call of a like batch acts on the first one's verdict
            d, u, c, w = rend.render_batch_ray(sc.c, dec, rd, ro, dev, sc.tsdf_volume, tsdf_bnds, 'color', gt_depth=gd)
            torch.cuda.synchronize()
    assert torch.isfinite(d).all() and torch.isfinite(c).all() and torch.isfinite(u).all()
    assert 0.02 < float((w != 1).float().mean()) < 0.9          # the band is hit, and not everywhere
    # These rays are a random subset of every pose's pixels: the order probe says so (Renderer._batch_is_incoherent), and a batch of
    # >= 65 536 such rays reads the CORNER-BLOCK copy of the volume (Renderer.tsdf_blocks = 'auto': one aligned 32-byte piece per
    # lookup, Engine.tsdf_blocks) instead of the volume as it stands.  Same values bit for bit -- and the same again rendered in sorted
    # order on top of that (round 4's path for such batches: rays are independent units) and with neither.
    assert rend._batch_is_incoherent(ro, rd, gd, sc.tsdf_volume, tsdf_bnds, wait=True)
    eng = rend._engine
    assert eng._tsdf_cb is not None and tuple(eng._tsdf_cb[1].shape) == tuple(sc.tsdf_volume.shape[2:][::-1]) + (8,)
    rend.sort_rays_min = 0                                       # nothing is looked at: the volume as it stands, the caller's order
    with torch.no_grad():
        d0, u0, c0, w0 = rend.render_batch_ray(sc.c, dec, rd, ro, dev, sc.tsdf_volume, tsdf_bnds, 'color', gt_depth=gd)
    rend.sort_rays_min = 65536
    assert torch.equal(d, d0) and torch.equal(u, u0) and torch.equal(c, c0) and torch.equal(w, w0)
    rend.sort_incoherent = True                                  # corner blocks AND sorted order
    with torch.no_grad():
        d2, u2, c2, w2 = rend.render_batch_ray(sc.c, dec, rd, ro, dev, sc.tsdf_volume, tsdf_bnds, 'color', gt_depth=gd)
    assert torch.equal(d, d2) and torch.equal(u, u2) and torch.equal(c, c2) and torch.equal(w, w2)
    rend.tsdf_blocks, rend.sort_incoherent = False, 'auto'       # no corner blocks: the sort takes over (round 4's path)
    assert rend._coherent_order(ro, rd, gd, sc.tsdf_volume, tsdf_bnds, wait=True) is not None
    with torch.no_grad():
        d3, u3, c3, w3 = rend.render_batch_ray(sc.c, dec, rd, ro, dev, sc.tsdf_volume, tsdf_bnds, 'color', gt_depth=gd)
    rend.tsdf_blocks = 'auto'
    assert torch.equal(d, d3) and torch.equal(u, u3) and torch.equal(c, c3) and torch.equal(w, w3)
    c2w = sc.default_c2w(yaw=0.7, pitch=-0.2)
    rp, dp = get_rays(sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, c2w, dev)
    assert not rend._batch_is_incoherent(rp.reshape(-1, 3)[:100000].contiguous(), dp.reshape(-1, 3)[:100000].contiguous(),      # pixel order: left alone
                                         sc.depth_image(c2w).reshape(-1)[:100000].contiguous(), sc.tsdf_volume, tsdf_bnds, wait=True)
    idx = torch.arange(0, n_rays, n_rays // check, device=dev)[:check]
    idx[0] = int(torch.argmax(gd))                               # keeps the batch-global far clamp of the subset equal
    cpu = {k: v.cpu() for k, v in sc.c.items()}
    od, ou, oc, ow = O.render_batch_ray(sd, cpu, rd[idx].cpu(), ro[idx].cpu(), sc.tsdf_volume.cpu(), sc.tsdf_bnds, sc.bound,
                                        'color', gd[idx].cpu(), 96, 32)
    assert int(((w[idx].cpu() == 1) != (ow == 1)).sum()) == 0    # no band-mask flips
    assert float((d[idx].cpu() - od).abs().max() / od.abs().max()) <= 1e-4
    assert float((c[idx].cpu() - oc).abs().max() / oc.abs().max()) <= 1e-4
    assert float((w[idx].cpu() - ow).abs().max()) <= 1e-4
