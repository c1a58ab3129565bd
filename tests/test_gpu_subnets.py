"""GPU (MI355X): the reference's public sub-modules called on their own -- ``decoders.low_decoder(p, c_grid)`` etc.
(MLP.forward, src/conv_onet/models/decoder.py:177-203) and ``decoders.mlp(p, occ, tsdf_volume, tsdf_bnds)``
(mlp_tsdf.forward, :240-258) -- through adfp_decode_single / adfp_attention_rows, against the vectors the REFERENCE's
own modules produced (tests/golden/mini_subnets.npz), in both math modes."""
import pytest
import torch

import attentive_dfprior_amd as A
from conftest import to_dev, assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = 1e-4


def make(mini):
    dec = A.DF()
    dec.load_state_dict(mini.sd)
    dec.bound = mini.bound
    for m in (dec.low_decoder, dec.high_decoder, dec.color_decoder):
        m.bound = mini.bound                                     # src/DF_Prior.py:192-194
    return dec.to(DEV)


@pytest.mark.parametrize('mode', ['f16x3', 'f32'])
def test_decoders_alone_vs_reference_golden(mini, mode, monkeypatch):
    monkeypatch.setenv('ADFP_MATH', mode)
    g = mini.golden('subnets')
    dec = make(mini)
    c = to_dev(mini.c, DEV)
    p = mini.query_points.to(DEV).unsqueeze(0)
    with torch.no_grad():
        for name in ('low', 'high', 'color'):
            out = getattr(dec, name + '_decoder')(p, c)
            assert tuple(out.shape) == g[name].shape and out.dtype == torch.float32
            assert_close(out, g[name], TOL, f'{name}_decoder(p f64)')
            out32 = getattr(dec, name + '_decoder')(p.float(), c)
            assert_close(out32, g[name + '_f32'], TOL, f'{name}_decoder(p f32)')


@pytest.mark.parametrize('mode', ['f16x3', 'f32'])
def test_attention_alone_vs_reference_golden(mini, mode, monkeypatch):
    monkeypatch.setenv('ADFP_MATH', mode)
    g = mini.golden('subnets')
    dec = make(mini)
    p = mini.query_points.to(DEV).unsqueeze(0)
    occ = torch.from_numpy(g['att_occ_in']).to(DEV)
    with torch.no_grad():
        fused, w = dec.mlp(p, occ, mini.tsdf_volume.to(DEV), mini.tsdf_bnds.to(DEV))
    assert tuple(fused.shape) == g['att_fused'].shape and tuple(w.shape) == g['att_w'].shape
    assert_close(fused, g['att_fused'], TOL, 'mlp_tsdf fused occupancy')
    assert_close(w, g['att_w'], TOL, 'mlp_tsdf attention weight')


def test_subnetwork_calls_are_inference_only(mini):
    """Gradients flow through the DF module; a sub-network on its own refuses to drop an autograd graph silently."""
    dec = make(mini)
    c = to_dev(mini.c, DEV)
    p = mini.query_points.to(DEV).unsqueeze(0)
    with pytest.raises(NotImplementedError):
        dec.low_decoder(p, c)                                    # parameters require grad and grad mode is on
    with torch.no_grad():
        dec.low_decoder(p, c)
