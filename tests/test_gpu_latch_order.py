"""GPU: a latch event that arrives BETWEEN two training iterations.

The status word of a DF module is written by the device at any time (an f16-range event of an earlier call) and read by the
host at the start of a call (DF.absorb_status): the network it names runs on the exact f32 kernels from then on.  The training
state a forward leaves for its backward (ReLU masks, layer inputs) is laid out from that latch, so the latch has to be read
BEFORE the layout: read after it, the state still had mask room for a network whose forward -- now exact -- never wrote it, and
the backward differentiated along whatever the slab held (one optimiser step on garbage per latch event).  Here the bit is set by
hand between two iterations; the gradients of the second must be the oracle's."""
import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import synthetic, _lib
from oracle import adfp_oracle as O
from conftest import make_cfg, assert_close, assert_close_scale

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('net', ['color', 'low', 'att'])
def test_latch_between_two_training_iterations(monkeypatch, net):
    monkeypatch.setenv('ADFP_MATH', 'f16x3')
    sc = synthetic.mini_scene()
    sd = O.random_state_dict(seed=3)
    ro, rd, gd, gc = [t.to(DEV) for t in synthetic.make_ray_batch(sc, 96, seed=5)]
    dec = A.DF()
    dec.load_state_dict(sd)
    dec.bound = sc.bound
    dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(32, 16), None, sc)
    tsdf, tb = sc.tsdf_volume.to(DEV), sc.tsdf_bnds.to(DEV)

    c_or = {k: v.clone().requires_grad_(True) for k, v in sc.c.items()}
    dec_or = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    o = O.render_batch_ray(dec_or, c_or, rd.cpu(), ro.cpu(), sc.tsdf_volume, sc.tsdf_bnds, sc.bound, 'color', gd.cpu(), 32, 16)
    O.mapper_loss(o[0], o[2], o[3], gd.cpu(), gc.cpu(), 'color', False).backward()

    def iteration():
        for p in dec.parameters():
            p.grad = None
        c = {k: v.to(DEV).clone().requires_grad_(True) for k, v in sc.c.items()}
        d, u, col, w = rend.render_batch_ray(c, dec, rd, ro, DEV, tsdf, tb, 'color', gt_depth=gd)
        O.mapper_loss(d, col, w, gd, gc, 'color', False).backward()
        return (d, u, col, w), c

    def check(out, c, what):
        for got, ref, name in zip(out, o, ('depth', 'uncertainty', 'color', 'weight')):
            assert_close(got, ref.detach(), 1e-4, f'{what}: {name}')
        for k, v in c.items():
            assert_close_scale(v.grad, c_or[k].grad, 2e-4, f'{what}: d/d {k}', flip_frac=2e-3)
        for name, p in dec.named_parameters():
            ref = dec_or[name].grad
            if ref is None or p.grad is None:
                continue
            assert bool(torch.isfinite(p.grad).all()), f'{what}: {name} gradient not finite'
            assert_close_scale(p.grad, ref, 1e-3, f'{what}: d/d {name}')

    out, c = iteration()
    check(out, c, 'before the latch')
    assert dec._exact_latch == set()
    torch.cuda.synchronize()
    dec.status_word()[0] = _lib.STATUS_RANGE_BITS[net]          # what a device-side range event leaves
    out, c = iteration()
    assert net in dec._exact_latch
    check(out, c, f'the iteration that finds {net} latched')
    out, c = iteration()
    check(out, c, 'the iteration after')
