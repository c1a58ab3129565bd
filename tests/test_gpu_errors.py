"""GPU: the error convention of the C ABI: 0 = ok, negative = argument / shape error
detected on the host before anything is launched, surfaced by the Python mirror as RuntimeError."""
import ctypes as C

import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import _lib, synthetic
from conftest import make_cfg

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _setup(ns=32, nf=16):
    sc = synthetic.mini_scene(device=DEV)
    dec = A.DF()
    dec.load_state_dict(synthetic.seeded_state_dict(1))
    dec.bound = sc.bound
    dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(ns, nf), None, sc)
    ro, rd, gd, _ = (t.to(DEV) for t in synthetic.make_ray_batch(sc, 50, seed=1))
    return sc, dec, rend, ro, rd, gd


def test_too_many_samples_is_rejected_not_truncated():
    sc, dec, rend, ro, rd, gd = _setup(ns=250, nf=16)            # 266 > ADFP_MAX_SAMPLES
    with torch.no_grad(), pytest.raises(RuntimeError, match='UNSUPPORTED|unsupported'):
        rend.render_batch_ray(sc.c, dec, rd, ro, DEV, sc.tsdf_volume, sc.tsdf_bnds.to(DEV), 'color', gt_depth=gd)


def test_workspace_too_small_and_null_pointers():
    sc, dec, rend, ro, rd, gd = _setup()
    L = _lib.lib()
    eng = rend._engine
    scene, keep = eng.scene(dec, sc.c, sc.tsdf_volume, sc.tsdf_bnds.to(DEV), sc.bound, 'color')
    N, S = ro.shape[0], 48
    a = _lib.AdfpRenderArgs()
    a.stage, a.n_rays, a.n_samples, a.n_surface = _lib.STAGE['color'], N, 32, 16
    a.rays_o, a.rays_d, a.gt_depth = ro.data_ptr(), rd.data_ptr(), gd.data_ptr()
    depth = torch.empty(N, dtype=torch.float64, device=DEV)
    unc = torch.empty(N, dtype=torch.float64, device=DEV)
    col = torch.empty(N, 3, device=DEV)
    w = torch.full((N, S), 7.0, device=DEV)
    a.depth, a.uncertainty, a.color, a.weight = depth.data_ptr(), unc.data_ptr(), col.data_ptr(), w.data_ptr()
    need = L.adfp_workspace_bytes(N * S)
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    a.workspace, a.workspace_bytes = ws.data_ptr(), need - 1
    st = _lib.current_stream(torch.device(DEV))
    rc = L.adfp_render_forward(C.byref(scene), C.byref(a), st)
    assert rc < 0 and 'WORKSPACE' in _lib.ERRORS.get(rc, '')
    torch.cuda.synchronize()
    assert bool((w == 7.0).all())                                   # nothing was launched
    a.workspace_bytes = need
    a.color = None
    assert L.adfp_render_forward(C.byref(scene), C.byref(a), st) < 0
    a.color = col.data_ptr()
    assert L.adfp_render_forward(C.byref(scene), C.byref(a), st) == 0
    torch.cuda.synchronize()
    assert bool(torch.isfinite(depth).all()) and not bool((w == 7.0).any())


def test_missing_weight_image_and_bad_stage():
    sc, dec, rend, ro, rd, gd = _setup()
    L = _lib.lib()
    scene, keep = rend._engine.scene(dec, sc.c, sc.tsdf_volume, sc.tsdf_bnds.to(DEV), sc.bound, 'color')
    pts = torch.rand(100, 3, device=DEV, dtype=torch.float64)
    ap = _lib.AdfpPoints()
    ap.mode, ap.n_points, ap.pts = _lib.PTS_F64, 100, pts.data_ptr()
    raw = torch.empty(100, 4, device=DEV)
    w = torch.empty(100, device=DEV)
    need = L.adfp_workspace_bytes(100)
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    st = _lib.current_stream(torch.device(DEV))
    assert L.adfp_eval_points(C.byref(scene), C.byref(ap), 7, 0, raw.data_ptr(), w.data_ptr(), ws.data_ptr(), need, st) < 0
    scene.h_color, scene.w_color = None, None                     # stage color without a colour decoder image
    assert L.adfp_eval_points(C.byref(scene), C.byref(ap), _lib.STAGE['color'], 0, raw.data_ptr(), w.data_ptr(),
                              ws.data_ptr(), need, st) < 0
    assert L.adfp_eval_points(C.byref(scene), C.byref(ap), _lib.STAGE['high'], 0, raw.data_ptr(), w.data_ptr(),
                              ws.data_ptr(), need, st) == 0
    torch.cuda.synchronize()


def test_a_failed_call_leaves_no_conversion_the_caches_vouch_for():
    """scene() defers the grid conversions and the weight-image packs of a call into ONE launch each and records them as current at
    once.  A call that dies in between -- here: the third grid has the wrong shape, after the first two were deferred -- must not
    leave cache entries for copies nobody wrote: the next, correct call renders what a fresh renderer renders, bit for bit."""
    sc, dec, rend, ro, rd, gd = _setup()
    tb = sc.tsdf_bnds.to(DEV)
    with torch.no_grad():
        want = A.Renderer(make_cfg(32, 16), None, sc).render_batch_ray(sc.c, dec, rd, ro, DEV, sc.tsdf_volume, tb, 'color', gt_depth=gd)
    fresh = {k: v.clone() for k, v in sc.c.items()}              # new tensors: nothing cached for them
    bad = dict(fresh)
    bad['grid_color'] = fresh['grid_color'][:, :16]               # [1,16,Z,Y,X]: grid_cl raises for the THIRD grid
    dec._packed.clear()
    rend._engine._grid_cache.clear()
    with torch.no_grad(), pytest.raises(RuntimeError, match='grid_color'):
        rend.render_batch_ray(bad, dec, rd, ro, DEV, sc.tsdf_volume, tb, 'color', gt_depth=gd)
    assert rend._engine._owed_relayouts is None and rend._engine._owed_packs is None
    with torch.no_grad():
        got = rend.render_batch_ray(fresh, dec, rd, ro, DEV, sc.tsdf_volume, tb, 'color', gt_depth=gd)
    for x, y in zip(got, want):
        assert torch.equal(x, y)
