"""GPU: a forward render captured into a HIP graph (torch.cuda.graph) replays bit-identically with new rays,
also with other allocations between the replays (the library's zero-fills are kernels, not memset nodes)."""
import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import synthetic
from conftest import make_cfg

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('stage,with_depth', [('color', True), ('high', False), ('low', True)])
def test_captured_render_replays_with_new_rays(stage, with_depth):
    sc = synthetic.mini_scene(device=DEV)
    dec = A.DF()
    dec.load_state_dict(synthetic.seeded_state_dict(3))
    dec.bound = sc.bound
    dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(32, 16), None, sc)
    bnds = sc.tsdf_bnds.to(DEV)
    batches = [tuple(t.to(DEV) for t in synthetic.make_ray_batch(sc, 300, seed=s)[:3]) for s in (1, 2, 3)]

    def render(o, d, z):
        return rend.render_batch_ray(sc.c, dec, d, o, DEV, sc.tsdf_volume, bnds, stage, gt_depth=z if with_depth else None)

    with torch.no_grad():
        refs = [render(*b) for b in batches]
        s_o, s_d, s_z = (t.clone() for t in batches[0])
        render(s_o, s_d, s_z)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = render(s_o, s_d, s_z)
        for k in (1, 2, 0, 1):
            for dst, src in zip((s_o, s_d, s_z), batches[k]):
                dst.copy_(src)
            g.replay()
            torch.cuda.synchronize()
            junk = (out[0] * 2).abs().max().item()           # allocations between the replays
            assert junk == junk
            for a, b in zip(out, refs[k]):
                assert torch.equal(a, b)
