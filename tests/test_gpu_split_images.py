"""GPU: the split weight images hold two operand layouts (H: 32x32x16 kernels, G: 16x16x32 kernels) and every forward keeps only
the part(s) it reads current (Engine.image_parts, include/adfp.h at adfp_pack_split_image).  After the parameters change, every
consumer must see the NEW weights whatever was rendered in between: a decoder that goes through inference in all three stages
and training forwards in all three stages, interleaved with optimiser steps, renders -- bit for bit -- what a fresh decoder
holding the same parameters renders."""
import itertools

import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import synthetic
from conftest import make_cfg

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _fresh(sd, sc):
    dec = A.DF()
    dec.load_state_dict({k: v.detach().clone() for k, v in sd.items()})
    dec.bound = sc.bound
    return dec.to(DEV)


def test_every_consumer_sees_the_current_weights(monkeypatch):
    monkeypatch.setenv('ADFP_MATH', 'f16x3')
    sc = synthetic.mini_scene(device=DEV)
    bnds = sc.tsdf_bnds.to(DEV)
    o, d, z = (t.to(DEV) for t in synthetic.make_ray_batch(sc, 200, seed=5)[:3])
    dec = _fresh(synthetic.seeded_state_dict(3), sc)
    rend = A.Renderer(make_cfg(32, 16), None, sc)
    opt = torch.optim.SGD(dec.parameters(), lr=1e-7)          # sum losses: large gradients; the weights must move, not explode
    stages = ('low', 'high', 'color')
    last = {}
    # (stage of the training step, then the stages rendered without grad) in every order that matters: the step leaves H parts
    # current, the renders need G for the fused low + colour launch / high / attention and H for a lone low decoder
    for k, (train_stage, order) in enumerate(itertools.product(stages, itertools.permutations(stages))):
        c = {key: v.detach().clone().requires_grad_(True) for key, v in sc.c.items()}
        opt.zero_grad(set_to_none=True)
        out = rend.render_batch_ray(c, dec, d, o, DEV, sc.tsdf_volume, bnds, train_stage, gt_depth=z)
        (out[0].float().sum() + out[2].sum()).backward()
        opt.step()
        twin = _fresh(dec.state_dict(), sc)
        assert all(torch.isfinite(p).all() for p in dec.parameters()) and not dec._exact_latch
        with torch.no_grad():
            for stage in order:
                got = rend.render_batch_ray(sc.c, dec, d, o, DEV, sc.tsdf_volume, bnds, stage, gt_depth=z)
                want = A.Renderer(make_cfg(32, 16), None, sc).render_batch_ray(sc.c, twin, d, o, DEV, sc.tsdf_volume, bnds, stage, gt_depth=z)
                for a, b in zip(got, want):
                    assert torch.isfinite(a).all()
                    assert torch.equal(a, b), f'step {k}: stage {stage} after a {train_stage} step rendered stale weights'
                # the step moved the weights enough to show: a stale image would reproduce the previous render
                assert stage not in last or not torch.equal(last[stage], got[0])
                last[stage] = got[0].clone()
        # and the NEXT training forward (H parts) after those renders equals the twin's
        a = rend.render_batch_ray(c, dec, d, o, DEV, sc.tsdf_volume, bnds, train_stage, gt_depth=z)
        b = A.Renderer(make_cfg(32, 16), None, sc).render_batch_ray(c, twin, d, o, DEV, sc.tsdf_volume, bnds, train_stage, gt_depth=z)
        for x, y in zip(a, b):
            assert torch.equal(x.detach(), y.detach()), f'step {k}: training forward of stage {train_stage} read stale weights'


def test_image_parts_rule():
    from attentive_dfprior_amd.engine import Engine
    eng = Engine()
    eng.inference_images = 'g'
    masks = {'masks_low': True, 'masks_high': True, 'masks_att': True, 'masks_color': True}
    for n in ('low', 'color'):                                   # the fused low + colour launch, inference and training
        assert eng.image_parts('color', n, set(), masks) == 'g' and eng.image_parts('color', n, set(), None) == 'g'
    for n in ('high', 'att'):
        assert eng.image_parts('color', n, set(), masks) == 'h' and eng.image_parts('color', n, set(), None) == 'g'
        assert eng.image_parts('high', n, set(), masks) == 'h' and eng.image_parts('high', n, set(), None) == 'g'
        assert eng.image_parts('color', n, set(), {}) == 'g'     # a training state without mask room: the inference kernel
    for st in ('low', 'high'):                                   # the low decoder on its own
        assert eng.image_parts(st, 'low', set(), None) == 'h' and eng.image_parts(st, 'low', set(), masks) == 'h'
    assert eng.image_parts('color', 'low', {'color'}, None) == 'h' and eng.image_parts('color', 'color', {'low'}, None) == 'h'
    assert eng.image_parts('color', 'low', set(), {}) == 'h' and eng.image_parts('color', 'low', set(), {'masks_low': True}) == 'h'
