"""GPU: seeded sweep over ray counts, sample counts, stages and depth modes against the oracle -- the ragged
corners of the tile / chunk logic (rays x samples not a multiple of 32, 2048, 256; S from 1 to 130; tiny and
mid-size batches; random scenes with their own weights)."""
import random

import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import synthetic
from oracle import adfp_oracle as O
from conftest import make_cfg, assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _cases():
    rng = random.Random(20240601)
    out = []
    for k in range(14):
        n_rays = rng.choice([1, 2, 3, 31, 33, 63, 65, 127, 255, 257, 500, 777, 1300])
        ns = rng.choice([1, 2, 7, 16, 31, 32, 33, 48, 64, 96, 100])
        nf = rng.choice([0, 1, 5, 16, 30, 32])
        stage = rng.choice(['low', 'high', 'color', 'color'])
        with_depth = rng.random() < 0.8
        out.append((k, n_rays, ns, nf, stage, with_depth))
    return out


@pytest.mark.parametrize('seed,n_rays,ns,nf,stage,with_depth', _cases())
def test_random_shapes_vs_oracle(seed, n_rays, ns, nf, stage, with_depth):
    scene = synthetic.mini_scene(seed=seed)
    sd = O.random_state_dict(seed=100 + seed)
    ro, rd, gd, _ = synthetic.make_ray_batch(scene, n_rays, seed=seed, zero_frac=0.15)
    dec = A.DF()
    dec.load_state_dict(sd)
    dec.bound = scene.bound
    dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(ns, nf), None, scene)
    c_dev = {k: v.to(DEV) for k, v in scene.c.items()}
    with torch.no_grad():
        d, u, c, w = rend.render_batch_ray(c_dev, dec, rd.to(DEV), ro.to(DEV), DEV, scene.tsdf_volume.to(DEV),
                                           scene.tsdf_bnds.to(DEV), stage, gt_depth=gd.to(DEV) if with_depth else None)
    od, ou, oc, ow, aux = O.render_batch_ray(sd, scene.c, rd, ro, scene.tsdf_volume, scene.tsdf_bnds, scene.bound, stage,
                                             gd if with_depth else None, ns, nf, return_aux=True)
    assert tuple(w.shape) == tuple(ow.shape) and d.dtype == od.dtype and c.dtype == oc.dtype
    # a sample whose TSDF value sits within float rounding of the band threshold may legitimately land on the
    # other side; none does on these seeds, so the comparison is unconditional
    assert_close(d, od, 1e-4, 'depth')
    assert_close(u, ou, 1e-4, 'uncertainty')
    assert_close(c, oc, 1e-4, 'color')
    assert_close(w, ow, 1e-4, 'weight')


@pytest.mark.parametrize('stage', ['high', 'color'])
def test_mesher_lattice_vs_oracle(stage):
    """The Mesher's query (src/utils/Mesher.py:286-326): a dense lattice over the bounding box plus a margin,
    through Renderer.eval_points in one call (the reference chunks it by 500 000 points; chunking is
    value-neutral), checked on a strided subset against the oracle."""
    scene = synthetic.mini_scene(seed=5)
    sd = O.random_state_dict(seed=105)
    dec = A.DF()
    dec.load_state_dict(sd)
    dec.bound = scene.bound
    dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(32, 16), None, scene)
    n = 96
    lo = scene.bound[:, 0] - 0.05 * (scene.bound[:, 1] - scene.bound[:, 0])
    hi = scene.bound[:, 1] + 0.05 * (scene.bound[:, 1] - scene.bound[:, 0])
    ax = [torch.linspace(float(lo[k]), float(hi[k]), n, dtype=torch.float64) for k in range(3)]
    pts = torch.stack(torch.meshgrid(*ax, indexing='ij'), -1).reshape(-1, 3)          # 884 736 points
    c_dev = {k: v.to(DEV) for k, v in scene.c.items()}
    with torch.no_grad():
        raw, w = rend.eval_points(pts.to(DEV), dec, scene.tsdf_volume.to(DEV), scene.tsdf_bnds.to(DEV), c_dev, stage, DEV)
    assert raw.shape == (pts.shape[0], 4) and w.shape == (pts.shape[0],)
    sub = torch.arange(0, pts.shape[0], 37)
    oraw, ow = O.eval_points(sd, pts[sub], scene.c, scene.tsdf_volume, scene.tsdf_bnds, scene.bound, stage)
    assert bool((oraw[:, 3] == 100).any()) and bool((ow != 1).any())                   # outside points and band points present
    assert_close(raw[sub.to(DEV)], oraw, 1e-4, 'raw')
    assert_close(w[sub.to(DEV)], ow, 1e-4, 'w')
