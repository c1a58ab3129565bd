"""GPU: the corner-block copy of the TSDF volume (adfp_relayout_tsdf, adfp_tsdf.corner_blocks).

The reference's volume is a permuted view with z fastest (/root/reference/get_tsdf.py:95-97); a trilinear lookup reads four 8-byte
column pieces that lie Z 4 and Y Z 4 bytes apart -- four 64-byte sectors per sample unless neighbouring lanes share them (rays in pixel
order do, a random ray batch does not: 175 B fetched per sample, profiles/r04_pmc_hbm_config5_random.csv).  The copy stores, per voxel,
the eight values a lookup with that lower corner blends, as one aligned 32-byte piece.  Checked here: the copy's contents against plain
indexing (clamped far faces), and the TSDF stage / whole renders through it against the strided path, bit for bit -- volumes with odd
sizes, points outside the volume (border clamp), the reference's permuted view and a contiguous [1,1,Z,Y,X] tensor."""
import ctypes as C

import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import synthetic, _lib
from oracle import adfp_oracle as O
from conftest import make_cfg, to_dev

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _volume(X, Y, Z, permuted, seed):
    g = torch.Generator().manual_seed(seed)
    phys = (torch.rand(X, Y, Z, generator=g) * 2.4 - 1.2).clamp(-1, 1).to(DEV)
    if permuted:
        return phys.reshape(1, 1, X, Y, Z).permute(0, 1, 4, 3, 2), phys              # the reference's view: strides (.., 1, Z, Y Z)
    return phys.permute(2, 1, 0).contiguous().reshape(1, 1, Z, Y, X), phys          # a plain contiguous [1,1,Z,Y,X] tensor


@pytest.mark.parametrize('dims,permuted', [((7, 5, 9), True), ((7, 5, 9), False), ((2, 2, 2), True), ((33, 18, 41), True)])
def test_corner_block_copy_holds_the_eight_blend_operands(dims, permuted):
    X, Y, Z = dims
    vol, phys = _volume(X, Y, Z, permuted, seed=X + Y + Z)
    cb = A.Renderer(make_cfg(), None, synthetic.mini_scene())._engine.tsdf_blocks(vol)
    assert tuple(cb.shape) == (X, Y, Z, 8)
    x, y, z = torch.meshgrid(torch.arange(X), torch.arange(Y), torch.arange(Z), indexing='ij')
    for k in range(8):
        dx, dy, dz = k & 1, (k >> 1) & 1, k >> 2
        want = phys[(x + dx).clamp(max=X - 1), (y + dy).clamp(max=Y - 1), (z + dz).clamp(max=Z - 1)]
        assert torch.equal(cb[..., k], want.to(DEV)), f'corner {k}'


def _tsdf_stage(eng, dec, sc, pts, blocks):
    """adfp_tsdf_stage on explicit f64 points: flags, in-band list (sorted), the list's inv_tsdf values and the raw trilerp."""
    L = _lib.lib()
    P = pts.shape[0]
    scn, keep = eng.scene(dec, to_dev(sc.c, DEV), sc.tsdf_volume.to(DEV) if not sc.tsdf_volume.is_cuda else sc.tsdf_volume,
                          sc.tsdf_bnds.to(DEV), sc.bound, 'color', tsdf_blocks=blocks)
    assert bool(scn.tsdf.corner_blocks) == blocks
    ap = _lib.AdfpPoints()
    ap.mode, ap.n_points, ap.pts = _lib.PTS_F64, P, pts.data_ptr()
    flags = torch.zeros((P,), dtype=torch.uint8, device=DEV)
    lst = torch.full((P,), -1, dtype=torch.int32, device=DEV)
    attu = torch.zeros((P,), dtype=torch.float32, device=DEV)
    cnt = torch.zeros((4,), dtype=torch.int32, device=DEV)
    _lib.check(L.adfp_tsdf_stage(C.byref(scn), C.byref(ap), _lib.ptr(flags), _lib.ptr(lst), _lib.ptr(attu), None, _lib.ptr(cnt),
                                 _lib.current_stream(DEV)), 'adfp_tsdf_stage')
    torch.cuda.synchronize()
    n = int(cnt[0])
    order = torch.argsort(lst[:n])
    return flags, lst[:n][order], attu[:n][order]


def test_tsdf_stage_through_the_copy_equals_the_strided_path_bit_for_bit():
    sc = synthetic.mini_scene()
    sc.tsdf_volume = sc.tsdf_volume.to(DEV)
    dec = A.DF()
    dec.load_state_dict(O.random_state_dict(seed=3))
    dec.bound = sc.bound
    dec = dec.to(DEV)
    eng = A.Renderer(make_cfg(), None, sc)._engine
    g = torch.Generator().manual_seed(1)
    lo, hi = sc.tsdf_bnds[:, 0], sc.tsdf_bnds[:, 1]
    pts = lo + (hi - lo) * (torch.rand(50000, 3, generator=g, dtype=torch.float64) * 1.3 - 0.15)     # 15 % beyond every face: border clamp
    pts[:8] = torch.stack([torch.where(torch.tensor([(k >> j) & 1 for j in range(3)], dtype=torch.bool), hi, lo) for k in range(8)])   # the corners
    pts = pts.to(DEV).contiguous()
    a = _tsdf_stage(eng, dec, sc, pts, blocks=False)
    b = _tsdf_stage(eng, dec, sc, pts, blocks=True)
    assert int(a[1].numel()) > 1000
    for x, y, what in zip(a, b, ('flags', 'in-band list', 'inv_tsdf of the list entries')):
        assert torch.equal(x, y), what


def test_render_through_the_copy_equals_the_strided_path_bit_for_bit(mini):
    sc = synthetic.mini_scene()
    dec = A.DF()
    dec.load_state_dict(O.random_state_dict(seed=3))
    dec.bound = sc.bound
    dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(32, 16), None, sc)
    ro, rd, gd, _ = [t.to(DEV) for t in synthetic.make_ray_batch(sc, 700, seed=5)]
    tsdf, tb, c = sc.tsdf_volume.to(DEV), sc.tsdf_bnds.to(DEV), to_dev(sc.c, DEV)
    with torch.no_grad():
        for stage in ('high', 'color'):
            plain = rend._engine.render_forward(dec, c, ro, rd, gd, tsdf, tb, sc.bound, stage, 32, 16)[:4]
            block = rend._engine.render_forward(dec, c, ro, rd, gd, tsdf, tb, sc.bound, stage, 32, 16, tsdf_blocks=True)[:4]
            for x, y in zip(plain, block):
                assert torch.equal(x, y), stage
    # the copy follows the volume: an in-place write (a fused frame, _version bumps) rebuilds it
    cb0 = rend._engine.tsdf_blocks(tsdf)
    assert rend._engine.tsdf_blocks(tsdf) is cb0
    tsdf.mul_(0.5)
    cb1 = rend._engine.tsdf_blocks(tsdf)
    assert cb1 is not cb0 and torch.equal(cb1[..., 0], tsdf[0, 0].permute(2, 1, 0))
