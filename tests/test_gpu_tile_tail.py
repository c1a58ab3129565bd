"""GPU: the chip-wide tile tail of the persistent inference kernels (claim_tile_pool, csrc/adfp_device.h).  Which workgroup computes a
tile must not matter: the fused low + colour launch with the pooled hand-out (tile counter given) equals the fixed split (NULL) bit
for bit at tile counts on both sides of the pooling threshold (6 rows of 256 x 12 tiles) and at ragged ends; and many back-to-back
frames complete (the hand-out spins on an LDS ring entry that another wave publishes)."""
import ctypes as C

import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import synthetic, _lib
from conftest import make_cfg

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _setup(n_rays, S=64):
    sc = synthetic.Scene('scene0050', H=120, W=160, fx=150.0, fy=150.0, cx=80.0, cy=60.0, device=DEV, grid_std_scale=20.0)
    dec = A.DF(); dec.load_state_dict(synthetic.seeded_state_dict(1)); dec.bound = sc.bound; dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(S - 16, 16), None, sc, ray_batch_size=10 ** 9)
    g = torch.Generator().manual_seed(n_rays)
    c2w = sc.default_c2w(yaw=0.2, pitch=-0.1)
    from attentive_dfprior_amd.common import get_rays
    ro, rd = get_rays(sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, c2w, DEV)
    pick = torch.randint(sc.H * sc.W, (n_rays,), generator=g).to(DEV)
    ro, rd = ro.reshape(-1, 3)[pick].contiguous(), rd.reshape(-1, 3)[pick].contiguous()
    gd = sc.depth_image(c2w).reshape(-1)[pick].contiguous()
    return sc, dec, rend, ro, rd, gd


@pytest.mark.parametrize('n_rays', [1, 300, 9215, 9216, 9217, 12289, 40000, 100003])
def test_pooled_hand_out_equals_the_fixed_split(n_rays, monkeypatch):
    monkeypatch.setenv('ADFP_MATH', 'f16x3')
    S = 64
    sc, dec, rend, ro, rd, gd = _setup(n_rays, S)
    eng = rend._engine
    tb = sc.tsdf_bnds.to(DEV)
    with torch.no_grad():
        out = eng.render_forward(dec, sc.c, ro, rd, gd, sc.tsdf_volume, tb, sc.bound, 'color', S - 16, 16, want_aux=True)
    aux = out[4]
    scn, keep = eng.scene(dec, sc.c, sc.tsdf_volume, tb, sc.bound, 'color')
    P = n_rays * S
    ap = _lib.AdfpPoints()
    ap.mode, ap.n_points = _lib.PTS_RAYS, P
    ap.rays_o, ap.rays_d, ap.z_vals, ap.S = ro.data_ptr(), rd.data_ptr(), aux['z_vals'].data_ptr(), S
    L = _lib.lib()
    st = _lib.current_stream(DEV)
    res = []
    for pooled in (False, True, True):
        raw = torch.full((P, 4), float('nan'), dtype=torch.float32, device=DEV)
        w = torch.full((P,), float('nan'), dtype=torch.float32, device=DEV)
        cnt = torch.full((1,), 12345, dtype=torch.int32, device=DEV)            # the entry zeroes it
        _lib.check(L.adfp_decode_stage(C.byref(scn), C.byref(ap), 3, _lib.ptr(raw), _lib.ptr(w), _lib.ptr(cnt) if pooled else None, st), 'decode')
        torch.cuda.synchronize()
        res.append((raw, w, int(cnt.item())))
    assert torch.equal(res[0][0].view(torch.int32), res[1][0].view(torch.int32)) and torch.equal(res[0][1].view(torch.int32), res[1][1].view(torch.int32))
    assert torch.equal(res[1][0].view(torch.int32), res[2][0].view(torch.int32))
    ntiles = (P + 31) // 32
    nwg = min((ntiles + 11) // 12, torch.cuda.get_device_properties(0).multi_processor_count)
    rows = (ntiles + nwg * 12 - 1) // (nwg * 12)
    if rows >= 6:
        assert res[1][2] > 0, 'a launch of this size hands its last rows out chip-wide'
    else:
        assert res[1][2] == 0, 'below six rows the split stays fixed'


def test_many_frames_complete(monkeypatch):
    """600 frames of 120 x 160 x 64 (9 600 tiles in the fused launch, ~1 600 in the in-band kernels) back to back."""
    monkeypatch.setenv('ADFP_MATH', 'f16x3')
    sc, dec, rend, ro, rd, gd = _setup(4)
    tb = sc.tsdf_bnds.to(DEV)
    c2w = sc.default_c2w(yaw=0.2, pitch=-0.1)
    depth = sc.depth_image(c2w)
    with torch.no_grad():
        first = rend.render_img(sc.c, dec, c2w, DEV, sc.tsdf_volume, tb, 'color', gt_depth=depth)
        for _ in range(600):
            out = rend.render_img(sc.c, dec, c2w, DEV, sc.tsdf_volume, tb, 'color', gt_depth=depth)
        torch.cuda.synchronize()
    for a, b in zip(first, out):
        assert torch.equal(a, b)
