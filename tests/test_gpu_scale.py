"""GPU: the HIP path at the feature scales the reference actually runs at, in BOTH arithmetic modes.

Every other parity test inflates the grids (std x 20..30, high grid x 100) so that the random decoders produce
non-trivial occupancies.  The reference initialises the grids N(0, 0.01) / N(0, 1e-4) / N(0, 0.01)
(src/DF_Prior.py:247-263) -- high-grid features are then f16 SUBNORMALS for the f16-split decoders -- and a
trained map has O(1..10) features and larger weights.  Both ends are checked here against the oracle, in
ADFP_MATH=f16x3 (default) and ADFP_MATH=f32, plus the range guard of the f16 split: operands beyond 65504 must
raise, not pass silently."""
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


def render_both(sc, sd, stage, n=400, ns=32, nf=16):
    ro, rd, gd, _ = synthetic.make_ray_batch(sc, n, seed=5)
    dec = A.DF()
    dec.load_state_dict(sd)
    dec.bound = sc.bound
    dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(ns, nf), None, sc)
    with torch.no_grad():
        out = rend.render_batch_ray(to_dev(sc.c, DEV), dec, rd.to(DEV), ro.to(DEV), DEV, sc.tsdf_volume.to(DEV),
                                    sc.tsdf_bnds.to(DEV), stage, gt_depth=gd.to(DEV))
    rend.check_overflow()
    ref = O.render_batch_ray(sd, sc.c, rd, ro, sc.tsdf_volume, sc.tsdf_bnds, sc.bound, stage, gd, ns, nf)
    return out, ref


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
    assert_close(u, ou, 5 * TOL, 'uncertainty')
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


def test_f16_range_guard_trips_on_features(monkeypatch):
    """A grid feature beyond the f16 range: the default mode must raise (sticky status word), the exact mode
    must render it."""
    sc = mini_at_scale(30.0, 100.0)
    sc.c['grid_color'] = sc.c['grid_color'].clone()
    sc.c['grid_color'][0, 3] = 1.0e5
    sd = O.random_state_dict(seed=3)
    monkeypatch.setenv('ADFP_MATH', 'f16x3')
    with pytest.raises(RuntimeError, match='65504'):
        render_both(sc, sd, 'color', n=64)
    _lib.check_status(sync=True)                                     # the flag was cleared by the raise
    monkeypatch.setenv('ADFP_MATH', 'f32')
    (d, u, c, w), (od, ou, oc, ow) = render_both(sc, sd, 'color', n=64)
    assert_close(d, od, TOL, 'depth (exact mode)')
    assert_close(c, oc, TOL, 'colour (exact mode)')


def test_f16_range_guard_trips_on_activations_and_weights(monkeypatch):
    monkeypatch.setenv('ADFP_MATH', 'f16x3')
    sc = mini_at_scale(30.0, 100.0)
    sd = O.random_state_dict(seed=3)
    big = {k: v.clone() for k, v in sd.items()}
    for k in big:                                                    # every operand < 65504, the hidden activations are not
        if k.startswith('color_decoder.pts_linears') and k.endswith('weight'):
            big[k] = big[k] * 300.0
    with pytest.raises(RuntimeError, match='65504'):
        render_both(sc, big, 'color', n=64)
    big = {k: v.clone() for k, v in sd.items()}
    big['mlp.pts_linears.1.weight'][0, 0] = 7.0e4                    # a weight the split cannot hold (pack time)
    with pytest.raises(RuntimeError, match='65504'):
        render_both(sc, big, 'color', n=64)
    # and the next clean call is clean
    (d, u, c, w), (od, ou, oc, ow) = render_both(sc, sd, 'color', n=64)
    assert_close(d, od, TOL, 'depth')
