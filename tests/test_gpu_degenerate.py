"""GPU: degenerate rays through the sampler (reference src/utils/Renderer.py:151-159, :203-221).

torch.max / torch.min / torch.clamp propagate NaN and torch.sort orders NaN after every number; the reference
therefore samples NaN along a whole ray whose slab test hits 0/0 (a zero direction component with the origin exactly
on that bound plane) and puts a NaN LAST when `lindisp` meets a zero sensor depth (inf * 0).  The HIP sampler must
reproduce the value AND the NaN pattern bit for bit, and the rendered ray must be NaN where the reference's is.
The scene's bound is made of binary fractions so that an f32 origin can sit exactly on an f64 bound plane."""
import numpy as np
import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import synthetic
from oracle import adfp_oracle as O
from conftest import make_cfg, to_dev, assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


class ExactScene(object):
    def __init__(self):
        self.bound = torch.tensor([[-1.0, 1.5], [-1.0, 1.25], [-0.75, 1.0]], dtype=torch.float64)
        self.c = synthetic.make_grids(self.bound, seed=4, std_scale=30.0)
        self.c['grid_high'] = self.c['grid_high'] * 100.0
        self.tsdf_volume, self.tsdf_bnds, (self.lo_in, self.hi_in) = synthetic.make_box_room_tsdf(
            self.bound, voxel=0.0625, inset=0.25, trunc_voxels=3.0)
        self.vol_bnds = self.tsdf_bnds
        self.H, self.W, self.fx, self.fy, self.cx, self.cy = 48, 64, 57.76, 57.76, 31.5, 23.5


def degenerate_rays():
    ctr = [0.25, 0.125, 0.125]
    ro = torch.tensor([ctr,
                       [-1.0, 0.125, 0.125],      # origin ON the lower x plane + zero x direction: 0/0 -> NaN ray
                       [1.5, 0.125, 0.125],       # ON the upper x plane + zero x direction
                       [0.25, 1.25, 0.125],       # ON the upper y plane, zero y direction
                       ctr, ctr, ctr, ctr,
                       [-1.0, 0.125, 0.125]],     # on the plane but with a non-zero direction: finite
                      dtype=torch.float32)
    rd = torch.tensor([[0.0, 0.3, -1.0],          # zero x, origin inside: +-inf planes, no NaN
                       [0.0, 0.3, -1.0],
                       [0.0, -0.2, -1.0],
                       [0.2, 0.0, -1.0],
                       [0.2, 0.0, -1.0],          # zero y, inside
                       [0.0, 0.0, -1.0],          # two zero components
                       [0.1, 0.2, -1.0],          # regular
                       [-0.3, 0.1, -1.0],
                       [0.5, 0.1, -1.0]], dtype=torch.float32)
    gd = torch.tensor([0.3, 0.3, 0.0, 0.25, 0.0, 0.2, 0.3, 0.0, 0.4], dtype=torch.float32)
    return ro, rd, gd


def same_bits(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    nan = np.isnan(b)
    return np.array_equal(np.isnan(a), nan) and np.array_equal(a[~nan], b[~nan])


@pytest.mark.parametrize('lindisp', [False, True])
@pytest.mark.parametrize('with_depth', [True, False])
def test_sampler_nan_pattern_bit_exact(lindisp, with_depth):
    sc = ExactScene()
    ro, rd, gd = degenerate_rays()
    ref = O.sample_z(ro, rd, gd if with_depth else None, sc.bound, 8, 4, lindisp=lindisp)
    if with_depth and not lindisp:
        assert torch.isnan(ref[1]).sum() == 8 and torch.isnan(ref[2]).sum() == 8 and torch.isnan(ref[3]).sum() == 8
        assert torch.isfinite(ref[[0, 4, 5, 6, 7, 8]]).all()
    if with_depth and lindisp:
        assert torch.isnan(ref[4, -1]) and torch.isfinite(ref[4, :-1]).all()        # inf * 0 in the last uniform sample
    sd = O.random_state_dict(seed=3)
    dec = A.DF()
    dec.load_state_dict(sd)
    dec.bound = sc.bound
    dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(8, 4, lindisp=lindisp), None, sc)
    with torch.no_grad():
        d, u, c, w, aux = rend._engine.render_forward(
            dec, to_dev(sc.c, DEV), ro.to(DEV), rd.to(DEV), gd.to(DEV) if with_depth else None, sc.tsdf_volume.to(DEV),
            sc.tsdf_bnds.to(DEV), sc.bound, 'color', 8, 4, lindisp=lindisp, want_aux=True)
    z = aux['z_vals'].cpu().numpy()
    r = ref.numpy()
    nan = np.isnan(r)
    assert np.array_equal(np.isnan(z), nan), (z, r)
    assert np.abs(z[~nan] - r[~nan]).max() <= 4e-16 * np.abs(r[~nan]).max()
    # rendered rays: NaN exactly where the reference's are; the others within tolerance
    od, ou, oc, ow = O.render_batch_ray(sd, sc.c, rd, ro, sc.tsdf_volume, sc.tsdf_bnds, sc.bound, 'color',
                                        gd if with_depth else None, 8, 4, lindisp=lindisp)
    bad = torch.isnan(od)
    assert torch.equal(torch.isnan(d.cpu()), bad) and torch.equal(torch.isnan(u.cpu()), torch.isnan(ou))
    assert torch.equal(torch.isnan(c.cpu()).any(-1), torch.isnan(oc).any(-1))
    ok = ~bad
    if ok.any():
        assert_close(d.cpu()[ok], od[ok], 1e-4, 'depth of the regular rays')
        assert_close(c.cpu()[ok], oc[ok], 1e-4, 'colour of the regular rays')
