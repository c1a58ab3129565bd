"""GPU: BASELINE.json configs[0] -- "Replica room0, 1k random rays, 32 samples/ray": the Mapper's own workload
(configs/df_prior.yaml:62 `pixels: 1000`, :94-95 `N_samples 32`, `N_surface 16`; rays by common.get_samples,
reference src/common.py:127-136 / src/Mapper.py:427-430) on the room0-sized synthetic scene of SURVEY.md section 8d
(grids initialised N(0, 0.01) / N(0, 1e-4) / N(0, 0.01) as src/DF_Prior.py:247-263, 785 MB TSDF, seed-0 decoders,
`torch.manual_seed(1)` pixel draw).  The HIP path (both math modes) against the oracle on the host cores: the three
stages forward at 1e-4, and the Mapper-loss gradients (src/Mapper.py:457-473) against the oracle's autograd."""
import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import common, synthetic
from oracle import adfp_oracle as O
from conftest import make_cfg, assert_close, assert_grad_tight, ReluCapture, assert_forced_decisions_are_boundary_units

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
N_RAYS, N_SAMPLES, N_SURFACE = 1000, 32, 16


@pytest.fixture(scope='module')
def cfg1():
    sc = synthetic.Scene('room0', device=DEV)                      # survey init scale: grid_std_scale = 1
    assert tuple(sc.c['grid_low'].shape[2:]) == (21, 28, 37) and tuple(sc.c['grid_high'].shape[2:]) == (43, 56, 74)
    assert tuple(sc.tsdf_volume.shape[2:]) == (451, 574, 758)
    sd = synthetic.seeded_state_dict(0)
    c2w = sc.default_c2w()
    depth = sc.depth_image(c2w)
    color = torch.rand((sc.H, sc.W, 3), generator=torch.Generator().manual_seed(0)).to(DEV)
    torch.manual_seed(1)
    ro, rd, gd, gc = common.get_samples(0, sc.H, 0, sc.W, N_RAYS, sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, c2w, depth, color, DEV)
    assert ro.shape == (N_RAYS, 3) and gd.shape == (N_RAYS,)
    cpu = dict(c={k: v.cpu() for k, v in sc.c.items()}, tsdf=sc.tsdf_volume.cpu(), tsdf_bnds=sc.tsdf_bnds.cpu(), bound=sc.bound.cpu(),
               ro=ro.detach().cpu(), rd=rd.detach().cpu(), gd=gd.cpu(), gc=gc.cpu())
    return sc, sd, (ro.detach(), rd.detach(), gd, gc), cpu


def make(sc, sd):
    dec = A.DF()
    dec.load_state_dict(sd)
    dec.bound = sc.bound
    return dec.to(DEV), A.Renderer(make_cfg(N_SAMPLES, N_SURFACE), None, sc)


@pytest.mark.parametrize('mode', ['f16x3', 'f32'])
@pytest.mark.parametrize('stage', O.STAGES)
def test_config1_forward_vs_oracle(cfg1, stage, mode, monkeypatch):
    monkeypatch.setenv('ADFP_MATH', mode)
    sc, sd, (ro, rd, gd, gc), cpu = cfg1
    dec, rend = make(sc, sd)
    with torch.no_grad():
        d, u, col, w = rend.render_batch_ray(sc.c, dec, rd, ro, DEV, sc.tsdf_volume, sc.tsdf_bnds.to(DEV), stage, gt_depth=gd)
        od, ou, oc, ow = O.render_batch_ray(sd, cpu['c'], cpu['rd'], cpu['ro'], cpu['tsdf'], cpu['tsdf_bnds'], cpu['bound'], stage,
                                            cpu['gd'], N_SAMPLES, N_SURFACE)
    assert tuple(w.shape) == (N_RAYS, N_SAMPLES + N_SURFACE, 1) and d.dtype == torch.float64
    assert_close(d, od, 1e-4, f'config 1 {stage} depth')
    assert_close(u, ou, 1e-4, f'config 1 {stage} uncertainty')
    assert_close(w, ow, 1e-4, f'config 1 {stage} attention weight')
    if stage == 'color':
        assert_close(col, oc, 1e-4, 'config 1 colour')
    # the band mask (w == 1 outside the TSDF band) must not flip anywhere
    assert int(((w.cpu().reshape(-1) == 1) != (ow.reshape(-1) == 1)).sum()) == 0


@pytest.mark.parametrize('mode', ['f16x3', 'f32'])
@pytest.mark.parametrize('stage,warm', [('low', False), ('high', True), ('color', False)])
def test_config1_mapper_gradients_vs_oracle_autograd(cfg1, stage, warm, mode, monkeypatch):
    """Every element of every grid and parameter gradient within conftest.TIGHT_GRAD_TOL (5e-5) x the tensor's scale of the
    oracle's autograd, differentiated along the ReLU decisions the kernels took (conftest.ReluCapture; the forced decisions
    differ from relu's own only on units within rounding of zero -- asserted)."""
    monkeypatch.setenv('ADFP_MATH', mode)
    sc, sd, (ro, rd, gd, gc), cpu = cfg1
    dec, rend = make(sc, sd)
    for p in dec.parameters():
        p.requires_grad_(True)
    cap = ReluCapture(rend)
    c = {k: v.clone().requires_grad_(True) for k, v in sc.c.items()}
    d, u, col, w = rend.render_batch_ray(c, dec, rd, ro, DEV, sc.tsdf_volume, sc.tsdf_bnds.to(DEV), stage, gt_depth=gd)
    loss = O.mapper_loss(d, col, w, gd, gc, stage, warm)
    loss.backward()
    c_or = {k: v.clone().requires_grad_(True) for k, v in cpu['c'].items()}
    sd_or = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    O.reset_relu_flips()
    od, ou, oc, ow = O.render_batch_ray(sd_or, c_or, cpu['rd'], cpu['ro'], cpu['tsdf'], cpu['tsdf_bnds'], cpu['bound'], stage,
                                        cpu['gd'], N_SAMPLES, N_SURFACE, relu_masks=cap.masks(stage))
    assert_forced_decisions_are_boundary_units(dict(O.RELU_FLIPS))
    loss_or = O.mapper_loss(od, oc, ow, cpu['gd'], cpu['gc'], stage, warm)
    loss_or.backward()
    assert abs(loss.item() - loss_or.item()) <= 1e-5 * abs(loss_or.item())
    for k in c:
        ref = c_or[k].grad if c_or[k].grad is not None else torch.zeros_like(c_or[k])
        assert_grad_tight(c[k].grad if c[k].grad is not None else torch.zeros_like(c[k]), ref, f'config 1 {stage} d/d {k}', mode)
    for name, p in dec.named_parameters():
        ref = sd_or[name].grad if sd_or[name].grad is not None else torch.zeros_like(sd_or[name])
        assert_grad_tight(p.grad if p.grad is not None else torch.zeros_like(p), ref, f'config 1 {stage} d/d {name}', mode)
