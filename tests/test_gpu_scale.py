"""GPU: the HIP path at the feature scales the reference actually runs at, in BOTH arithmetic modes.

Every other parity test inflates the grids (std x 20..30, high grid x 100) so that the random decoders produce
non-trivial occupancies.  The reference initialises the grids N(0, 0.01) / N(0, 1e-4) / N(0, 0.01)
(src/DF_Prior.py:247-263) -- high-grid features are then f16 SUBNORMALS for the f16-split decoders -- and a
trained map has O(1..10) features and larger weights.  Both ends are checked here against the oracle, in
ADFP_MATH=f16x3 (default) and ADFP_MATH=f32, plus the range guard of the f16 split: a call that meets an operand beyond
65504 must hand out CORRECT values (the device-side f32 repair, csrc/adfp_fallback.h) without raising, and the network that
tripped must run on the exact kernels from the next call on."""
import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import synthetic, _lib
from oracle import adfp_oracle as O
from conftest import make_cfg, to_dev, assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = 1e-4


def mini_at_scale(grid_std_scale, high_factor):
    sc = synthetic.mini_scene(grid_std_scale=grid_std_scale)      # mini_scene() multiplies the high grid by 100
    sc.c['grid_high'] = sc.c['grid_high'] / 100.0 * high_factor
    return sc


def render_both(sc, sd, stage, n=400, ns=32, nf=16, expect_latched=frozenset(), calls=1):
    ro, rd, gd, _ = synthetic.make_ray_batch(sc, n, seed=5)
    dec = A.DF()
    dec.load_state_dict(sd)
    dec.bound = sc.bound
    dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(ns, nf), None, sc)
    outs = []
    with torch.no_grad():
        for _ in range(calls):
            outs.append(rend.render_batch_ray(to_dev(sc.c, DEV), dec, rd.to(DEV), ro.to(DEV), DEV, sc.tsdf_volume.to(DEV),
                                              sc.tsdf_bnds.to(DEV), stage, gt_depth=gd.to(DEV)))
    latched = rend.check_overflow(DEV, dec)
    assert latched == set(expect_latched), f'latched to exact: {latched}, expected {set(expect_latched)}'
    ref = O.render_batch_ray(sd, sc.c, rd, ro, sc.tsdf_volume, sc.tsdf_bnds, sc.bound, stage, gd, ns, nf)
    return (outs[0] if calls == 1 else outs), ref


@pytest.mark.parametrize('math', ['f16x3', 'f32'])
@pytest.mark.parametrize('stage', ['high', 'color'])
def test_reference_init_scale(monkeypatch, math, stage):
    """grids exactly as src/DF_Prior.py:247-263 initialises them: std 0.01 / 1e-4 / 0.01."""
    monkeypatch.setenv('ADFP_MATH', math)
    sc = mini_at_scale(1.0, 1.0)
    assert float(sc.c['grid_high'].abs().max()) < 1e-3              # f16-subnormal territory for the hi parts
    sd = O.random_state_dict(seed=3)
    (d, u, c, w), (od, ou, oc, ow) = render_both(sc, sd, stage)
    assert int(((w.cpu() == 1) != (ow == 1)).sum()) == 0
    assert_close(d, od, TOL, 'depth')
    assert_close(u, ou, TOL, 'uncertainty')
    assert_close(w, ow, TOL, 'attention weight')
    if stage == 'color':
        assert_close(c, oc, TOL, 'colour')


@pytest.mark.parametrize('math', ['f16x3', 'f32'])
def test_trained_scale(monkeypatch, math):
    """features up to |c| ~ 10 and decoder weights several times the init scale: hidden activations in the
    hundreds -- still far inside the f16 range, and the split must stay fp32-grade."""
    monkeypatch.setenv('ADFP_MATH', math)
    sc = mini_at_scale(250.0, 100.0)                               # std 2.5 on all three grids
    assert 8.0 < float(sc.c['grid_color'].abs().max()) < 20.0
    sd = O.random_state_dict(seed=5)
    for k in sd:
        if 'fc_c' in k and k.endswith('weight'):
            sd[k] = sd[k] * 8.0
        elif 'pts_linears' in k and k.endswith('weight') and not k.startswith('mlp'):
            sd[k] = sd[k] * 2.0
    for name in ('low', 'high'):                                     # keep the occupancy in sigmoid's live range
        sd[f'{name}_decoder.output_linear.weight'] = sd[f'{name}_decoder.output_linear.weight'] / 2000.0
    (d, u, c, w), (od, ou, oc, ow) = render_both(sc, sd, 'color')
    assert float(oc.abs().max()) > 50.0                              # the activations really are large
    assert_close(d, od, TOL, 'depth')
    assert_close(c, oc, TOL, 'colour')
    assert_close(w, ow, TOL, 'attention weight')


def check(out, ref, what, extreme_logits=False):
    """extreme_logits: the case puts a weight of 7e4 into the attention MLP.  Its two logits are then O(1e4) numbers whose float32
    rounding (ulp 1e-3, in the oracle as much as in the kernel: both sum in float32, in different orders) IS the relative error
    of the smaller softmax output, so that output is held to 1e-5 of the weight's range [0, 1] there instead of to a ratio."""
    d, u, c, w = out
    od, ou, oc, ow = ref
    assert_close(d, od, TOL, f'depth ({what})')
    assert_close(c, oc, TOL, f'colour ({what})')
    if extreme_logits:
        from conftest import assert_close_scale
        assert_close_scale(w, ow, 1e-5, f'attention weight ({what})')
    else:
        assert_close(w, ow, TOL, f'attention weight ({what})')


def test_f16_range_violation_on_features_is_repaired(monkeypatch):
    """A grid feature beyond the f16 range.  f16x3 mode: the tripping call itself returns correct values (f32 repair on the
    device, no exception), the colour decoder is exact afterwards and the second call (exact kernel) agrees too; f32 mode
    renders it directly."""
    sc = mini_at_scale(30.0, 100.0)
    sc.c['grid_color'] = sc.c['grid_color'].clone()
    sc.c['grid_color'][0, 3] = 1.0e5
    sd = O.random_state_dict(seed=3)
    monkeypatch.setenv('ADFP_MATH', 'f16x3')
    (first, second), ref = render_both(sc, sd, 'color', n=64, expect_latched={'color'}, calls=2)
    check(first, ref, 'the call that tripped')
    check(second, ref, 'the next call, colour decoder latched to exact')
    monkeypatch.setenv('ADFP_MATH', 'f32')
    out, ref = render_both(sc, sd, 'color', n=64)
    check(out, ref, 'exact mode')


def test_f16_range_violation_on_activations_and_weights_is_repaired(monkeypatch):
    monkeypatch.setenv('ADFP_MATH', 'f16x3')
    sc = mini_at_scale(30.0, 100.0)
    sd = O.random_state_dict(seed=3)
    big = {k: v.clone() for k, v in sd.items()}
    for k in big:                                                    # every operand < 65504, the hidden activations are not
        if k.startswith('color_decoder.pts_linears') and k.endswith('weight'):
            big[k] = big[k] * 300.0
    for k in big:                                                    # keep the oracle's colour finite in f32
        if k.startswith('color_decoder.output_linear'):
            big[k] = big[k] * 1e-6
    out, ref = render_both(sc, big, 'color', n=64, expect_latched={'color'})
    assert float(ref[2].abs().max()) > 0
    check(out, ref, 'hidden activations beyond the f16 range')
    big = {k: v.clone() for k, v in sd.items()}
    big['mlp.pts_linears.1.weight'][0, 0] = 7.0e4                    # a weight the split cannot hold (found at pack time)
    out, ref = render_both(sc, big, 'color', n=64, expect_latched={'att'})
    check(out, ref, 'a weight beyond the f16 range', extreme_logits=True)
    big = {k: v.clone() for k, v in sd.items()}
    big['low_decoder.fc_c.2.weight'][5, 7] = -9.0e4                  # the low decoder: feeds the in-band list too (high + low)
    for stage in ('low', 'high', 'color'):
        out, ref = render_both(sc, big, stage, n=64, expect_latched={'low'})
        assert_close(out[0], ref[0], TOL, f'depth (stage {stage}, low decoder repaired)')
        assert_close(out[3], ref[3], TOL, f'attention weight (stage {stage}, low decoder repaired)')
    # a clean network stays on the f16-split kernels
    out, ref = render_both(sc, sd, 'color', n=64, expect_latched=set())
    check(out, ref, 'clean')


def test_f16_range_violation_in_training_gives_correct_outputs_and_zero_gradients(monkeypatch):
    """Training forward: outputs repaired like in inference; the state it left (ReLU masks, layer inputs) is not valid, so the
    backward of THAT call returns zero gradients (never garbage); the next iteration runs the latched network exactly and its
    gradients agree with the oracle's autograd."""
    from conftest import assert_close_scale
    monkeypatch.setenv('ADFP_MATH', 'f16x3')
    sc = mini_at_scale(30.0, 100.0)
    sd = O.random_state_dict(seed=3)
    sd['color_decoder.fc_c.1.weight'][2, 4] = 8.0e4
    ro, rd, gd, gc = [t.to(DEV) for t in synthetic.make_ray_batch(sc, 96, seed=5)]
    dec = A.DF()
    dec.load_state_dict(sd)
    dec.bound = sc.bound
    dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(32, 16), None, sc)
    tsdf, tb = sc.tsdf_volume.to(DEV), sc.tsdf_bnds.to(DEV)
    ref_out = O.render_batch_ray(sd, sc.c, rd.cpu(), ro.cpu(), sc.tsdf_volume, sc.tsdf_bnds, sc.bound, 'color', gd.cpu(), 32, 16)
    c_or = {k: v.clone().requires_grad_(True) for k, v in sc.c.items()}
    o2 = O.render_batch_ray(sd, c_or, rd.cpu(), ro.cpu(), sc.tsdf_volume, sc.tsdf_bnds, sc.bound, 'color', gd.cpu(), 32, 16)
    O.mapper_loss(o2[0], o2[2], o2[3], gd.cpu(), gc.cpu(), 'color', False).backward()
    for it in range(2):
        c = {k: v.to(DEV).clone().requires_grad_(True) for k, v in sc.c.items()}
        d, u, col, w = rend.render_batch_ray(c, dec, rd, ro, DEV, tsdf, tb, 'color', gt_depth=gd)
        check((d, u, col, w), ref_out, f'training forward, iteration {it}')
        O.mapper_loss(d, col, w, gd, gc, 'color', False).backward()
        if it == 0:
            for k, v in c.items():
                assert v.grad is None or float(v.grad.abs().max()) == 0.0, f'{k}: a repaired call must return zero gradients'
            # ... for the decoder parameters too: on the autograd path torch.optim.Adam would apply whatever arrives here
            for name, p in dec.named_parameters():
                assert p.grad is None or (bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) == 0.0), \
                    f'{name}: a repaired call must return zero (finite) parameter gradients'
            for p in dec.parameters():
                p.grad = None
        else:
            assert rend.check_overflow(DEV, dec) == {'color'}
            for k, v in c.items():
                assert_close_scale(v.grad, c_or[k].grad, 2e-4, f'd/d {k} after the latch', flip_frac=2e-3)


def test_fused_iteration_skips_the_step_of_a_repaired_forward(monkeypatch):
    """mapping.MapperIteration (graph replay).  A replay whose forward leaves the f16 range (here: a grid feature that grew to
    1e5 between two iterations) repairs its outputs, returns zero gradients and changes NOTHING -- parameters, grids, Adam
    moments, step counters; the following iterations run the latched network on the exact kernels and train."""
    from attentive_dfprior_amd import mapping
    monkeypatch.setenv('ADFP_MATH', 'f16x3')
    sc = mini_at_scale(30.0, 100.0)
    sd = O.random_state_dict(seed=3)
    rays = [t.to(DEV) for t in synthetic.make_ray_batch(sc, 256, seed=5)]
    dec = A.DF()
    dec.load_state_dict(sd)
    dec.bound = sc.bound
    dec = dec.to(DEV)
    for p in list(dec.low_decoder.parameters()) + list(dec.high_decoder.parameters()):
        p.requires_grad_(False)
    rend = A.Renderer(make_cfg(32, 16), None, sc)
    grids = {k: v.clone().to(DEV) for k, v in sc.c.items()}
    lr = {st: dict(low=0.005, high=0.005, color=0.005, decoders=0.005, mlp=0.005) for st in ('low', 'high', 'color')}
    it = mapping.MapperIteration(rend, dec, grids, None, sc.tsdf_volume.to(DEV), sc.tsdf_bnds.to(DEV), lr, use_graph=True)
    float(it.step(*rays, 'color'))                                   # captures the f16-split graph, clean
    assert dec._exact_latch == set() and int(it.step_count.max()) == 1
    grids['grid_color'][0, 3] = 1.0e5                                # in place, like a diverging optimiser would
    torch.autograd.graph.increment_version(grids['grid_color'])
    before = {k: v.clone() for k, v in grids.items()}
    pbefore = {n: p.detach().clone() for n, p in dec.named_parameters()}
    mbefore = {k: (m.clone(), v.clone()) for k, (m, v) in list(it.gstate.items()) + list(it.fstate.items())}
    it.step(*rays, 'color')                                          # replay: trips, repairs, skips
    torch.cuda.synchronize()
    assert int(it.step_count.max()) == 1 and int(it.step_count.min()) == 1
    for k in grids:
        assert torch.equal(grids[k], before[k]), f'{k} moved in the iteration whose forward was repaired'
    for n, p in dec.named_parameters():
        assert torch.equal(p.detach(), pbefore[n]), n
    for k, (m, v) in list(it.gstate.items()) + list(it.fstate.items()):
        assert torch.equal(m, mbefore[k][0]) and torch.equal(v, mbefore[k][1]), f'Adam moments of {k} moved'
    l3 = float(it.step(*rays, 'color'))                              # absorbs the status word, captures the exact-colour graph, steps
    assert dec._exact_latch == {'color'}
    assert int(it.step_count.max()) == 2
    assert any(not torch.equal(grids[k], before[k]) for k in grids)
    l4 = float(it.step(*rays, 'color'))
    assert torch.isfinite(torch.tensor([l3, l4])).all() and l4 < l3
