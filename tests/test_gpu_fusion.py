"""GPU: TSDF fusion kernel.  PINNED since round 4 against the REFERENCE'S OWN KERNEL: the CUDA C string
of src/fusion.py:69-142 is plain CUDA C; oracle/build_ref_fusion.py compiles it with hipcc from where it lies under /root/reference
into oracle/_ref/ (twice: -ffp-contract=off = every operation rounded as written, and the compiler's default contraction like
nvcc's -fmad under PyCUDA) and test_integrate_matches_the_reference_kernel runs it on the MI355X beside adfp_tsdf_integrate, with
the launch geometry of src/fusion.py:146-154 / :226-251 -- bit for bit against the as-written build, float32 rounding against the
contracted one; the numpy restatement (oracle.tsdf_integrate_np) is held to the same kernel there.  Two further checks: (1) bit
for bit against that numpy restatement; (2) against an INDEPENDENT float64 fusion written from the method, not from the kernel
(integer lattice coordinates, float64 projection): away from pixel-rounding ties and the truncation edge the two must agree to
float32 rounding."""
import numpy as np
import pytest
import torch

from attentive_dfprior_amd import synthetic
from attentive_dfprior_amd.fusion import TSDFVolume
from oracle import adfp_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def frames(sc, n):
    out = []
    for k in range(n):
        c2w = sc.default_c2w(offset=(0.05 * k, -0.04 * k, 0.02), yaw=0.9 * k, pitch=0.1 * k - 0.1)
        depth = sc.depth_image(c2w, zero_band=0.08).cpu().numpy().astype(np.float32)
        g = np.random.default_rng(k)
        color = g.integers(0, 256, size=(sc.H, sc.W, 3)).astype(np.uint8)
        pose = c2w.cpu().numpy().astype(np.float64).copy()
        pose[:3, 1] *= -1.0                       # OpenGL -> OpenCV camera, get_tsdf.py:79-80
        pose[:3, 2] *= -1.0
        K = np.array([[sc.fx, 0, sc.cx], [0, sc.fy, sc.cy], [0, 0, 1]], dtype=np.float64)
        out.append((color, depth, K, pose))
    return out


@pytest.mark.parametrize('voxel', [0.04, 0.0062])       # 40x40x32 and 259x259x207 = 13.9 M (> 2^23: float index rounding)
def test_integrate_matches_cuda_kernel_restatement(voxel):
    sc = synthetic.mini_scene(device=DEV)
    vol = TSDFVolume(sc.bound.numpy(), voxel, device=DEV)
    t = np.full(tuple(vol._vol_dim), -1.0, np.float32)
    w = np.zeros_like(t)
    c = np.zeros_like(t)
    for color, depth, K, pose in frames(sc, 3):
        vol.integrate(color, depth, K, pose, obs_weight=1.0)
        packed = np.floor(color[..., 2].astype(np.float32) * 65536 + color[..., 1].astype(np.float32) * 256 + color[..., 0].astype(np.float32))
        t, w, c = O.tsdf_integrate_np(t, w, c, vol._vol_origin, vol._voxel_size, K, pose, packed, depth, vol._trunc_margin, 1.0)
    gt, gc, bnds = vol.get_volume()
    assert (vol._weight.cpu().numpy() > 0).mean() > 0.05
    assert np.array_equal(vol._weight.cpu().numpy(), w)
    bad = gt != t
    assert not bad.any(), (int(bad.sum()), float(np.abs(gt - t).max()), gt[bad][:5], t[bad][:5], w[bad][:5])
    assert np.array_equal(gc, c)
    assert gt.min() >= -1.0 and gt.max() <= 1.0


def _ref_lib(name):
    import ctypes as C
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_ref', name)
    if not os.path.exists(path):
        pytest.skip(f'{path} has not been built (oracle/build_ref_fusion.py needs /root/reference)')
    lib = C.CDLL(path)
    lib.ref_fusion_integrate.restype = C.c_int
    lib.ref_fusion_integrate.argtypes = [C.c_void_p] * 10 + [C.c_int] * 5 + [C.c_void_p]
    return lib


def _reference_launch_geometry(n_vox, threads=1024, max_grid=(2147483647, 65535, 65535)):
    """src/fusion.py:146-154 (gpu_dev.MAX_THREADS_PER_BLOCK = 1024; MAX_GRID_DIM_* as HIP reports them for gfx950)"""
    n_blocks = int(np.ceil(float(n_vox) / float(threads)))
    gx = min(max_grid[0], int(np.floor(np.cbrt(n_blocks))))
    gy = min(max_grid[1], int(np.floor(np.sqrt(n_blocks / gx))))
    gz = min(max_grid[2], int(np.ceil(float(n_blocks) / float(gx * gy))))
    loops = int(np.ceil(float(n_vox) / float(gx * gy * gz * threads)))
    return gx, gy, gz, loops


@pytest.mark.parametrize('voxel', [0.04, 0.0062])       # 40x40x32 and 259x259x207 = 13.9 M voxels (> 2^23: the float index decomposition rounds)
def test_integrate_matches_the_reference_kernel(voxel):
    """adfp_tsdf_integrate (and the numpy restatement) against the reference's CUDA kernel string compiled by hipcc and launched
    like PyCUDA launches it: three frames into one volume; tsdf, weight and packed-colour volumes bit for bit."""
    from attentive_dfprior_amd import _lib
    exact, contract = _ref_lib('libref_fusion_exact.so'), _ref_lib('libref_fusion_contract.so')
    sc = synthetic.mini_scene(device=DEV)
    vol = TSDFVolume(sc.bound.numpy(), voxel, device=DEV)
    dims = tuple(int(v) for v in vol._vol_dim)
    n = int(np.prod(dims))
    gx, gy, gz, loops = _reference_launch_geometry(n)

    def fresh():      # + 1 element: the reference's bound check is `voxel_idx > N`, so thread N writes one element past the volume
        t = torch.full((n + 1,), -1.0, device=DEV)
        return t, torch.zeros(n + 1, device=DEV), torch.zeros(n + 1, device=DEV)
    state = {'exact': fresh(), 'contract': fresh()}
    t_np = np.full(dims, -1.0, np.float32)
    w_np, c_np = np.zeros_like(t_np), np.zeros_like(t_np)
    st = _lib.current_stream(torch.device(DEV))
    for color, depth, K, pose in frames(sc, 3):
        vol.integrate(color, depth, K, pose, obs_weight=1.0)
        packed = np.floor(color[..., 2].astype(np.float32) * 65536 + color[..., 1].astype(np.float32) * 256 + color[..., 0].astype(np.float32))
        t_np, w_np, c_np = O.tsdf_integrate_np(t_np, w_np, c_np, vol._vol_origin, vol._voxel_size, K, pose, packed, depth, vol._trunc_margin, 1.0)
        im_h, im_w = depth.shape
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32).reshape(-1)).to(DEV)            # noqa: E731
        args = [dev(vol._vol_dim.astype(np.float32)), dev(vol._vol_origin), dev(K), dev(pose),
                dev(np.array([[k, vol._voxel_size, im_h, im_w, vol._trunc_margin, 1.0] for k in range(loops)], np.float32)), dev(packed), dev(depth)]
        for name, lib in (('exact', exact), ('contract', contract)):
            t, w, c = state[name]
            rc = lib.ref_fusion_integrate(t.data_ptr(), w.data_ptr(), c.data_ptr(), *[a.data_ptr() for a in args], loops, gx, gy, gz, 1024, st)
            assert rc == 0
        torch.cuda.synchronize()
    ours_t, ours_c, _ = vol.get_volume()
    ours_w = vol._weight.cpu().numpy()
    rt, rw, rc_ = (x[:n].cpu().numpy().reshape(dims) for x in state['exact'])
    assert (rw > 0).mean() > 0.05
    # as written (no contraction): the product kernel and the numpy restatement ARE the reference's arithmetic
    assert np.array_equal(ours_w, rw) and np.array_equal(ours_t, rt) and np.array_equal(ours_c, rc_)
    assert np.array_equal(w_np, rw) and np.array_equal(t_np, rt) and np.array_equal(c_np, rc_)
    # with the compiler free to contract mul + add into fma (what nvcc does under PyCUDA by default): the projection moves by an
    # ulp, so a voxel whose pixel centre or truncation edge sits on a rounding tie may take the other branch; everywhere else the
    # two agree to float32 rounding
    ct, cw, cc = (x[:n].cpu().numpy().reshape(dims) for x in state['contract'])
    same = cw == rw
    assert same.mean() > 0.999, same.mean()
    assert np.abs(ct[same] - rt[same]).max() <= 2e-5          # one ulp of the camera-space depth (~2.4e-7 at 3 m) over the truncation margin (0.031 m)
    assert (cc[same] == rc_[same]).mean() > 0.999


def fuse_f64(tsdf, weight, origin, voxel, K, pose, depth, trunc, obs_w=1.0):
    """KinectFusion-style integration of one depth frame in float64, written from the method: project every voxel centre with the
    inverse pose, nearest pixel, truncated signed distance along the optical axis, weighted running mean.  Returns the new
    volumes and a mask of the voxels whose outcome does not hinge on a rounding tie (pixel centre within 1e-3 px of x.5, image
    border, truncation edge or camera plane within 1e-4)."""
    nx, ny, nz = tsdf.shape
    ix, iy, iz = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing='ij')
    p = np.stack([origin[0] + ix * float(voxel), origin[1] + iy * float(voxel), origin[2] + iz * float(voxel)], -1).astype(np.float64)
    R, t = np.asarray(pose, np.float64)[:3, :3], np.asarray(pose, np.float64)[:3, 3]
    cam = (p - t) @ R                                              # R^T (p - t)
    with np.errstate(divide='ignore', invalid='ignore'):
        u = K[0, 0] * cam[..., 0] / cam[..., 2] + K[0, 2]
        v = K[1, 1] * cam[..., 1] / cam[..., 2] + K[1, 2]
    pu = np.where(u >= 0, np.floor(u + 0.5), np.ceil(u - 0.5))
    pv = np.where(v >= 0, np.floor(v + 0.5), np.ceil(v - 0.5))
    h, w = depth.shape
    inside = np.isfinite(pu) & np.isfinite(pv) & (pu >= 0) & (pu < w) & (pv >= 0) & (pv < h) & (cam[..., 2] >= 0)
    z = np.zeros_like(u)
    z[inside] = depth[pv[inside].astype(int), pu[inside].astype(int)]
    ahead = z - cam[..., 2]
    hit = inside & (z != 0) & (ahead >= -trunc)
    sd = np.minimum(1.0, ahead / trunc)
    w_new = weight + obs_w
    out_t, out_w = tsdf.astype(np.float64).copy(), weight.astype(np.float64).copy()
    out_t[hit] = (out_t[hit] * weight[hit] + obs_w * sd[hit]) / w_new[hit]
    out_w[hit] = w_new[hit]
    with np.errstate(invalid='ignore'):
        fu, fv = np.abs(u - np.floor(u) - 0.5), np.abs(v - np.floor(v) - 0.5)
        robust = np.isfinite(u) & np.isfinite(v) & (fu > 1e-3) & (fv > 1e-3) & (np.abs(cam[..., 2]) > 1e-4) & (np.abs(ahead + trunc) > 1e-4) \
            & (np.abs(u + 0.5) > 1e-3) & (np.abs(u - (w - 0.5)) > 1e-3) & (np.abs(v + 0.5) > 1e-3) & (np.abs(v - (h - 0.5)) > 1e-3)
    return out_t, out_w, robust


def test_integrate_against_an_independent_float64_fusion():
    sc = synthetic.mini_scene(device=DEV)
    vol = TSDFVolume(sc.bound.numpy(), 0.04, device=DEV)
    t = np.full(tuple(vol._vol_dim), -1.0, np.float64)
    w = np.zeros_like(t)
    robust = np.ones(t.shape, bool)
    for color, depth, K, pose in frames(sc, 3):
        vol.integrate(color, depth, K, pose, obs_weight=1.0)
        t, w, ok = fuse_f64(t, w, vol._vol_origin.astype(np.float64), vol._voxel_size, K, pose, depth.astype(np.float64), float(np.float32(vol._trunc_margin)))
        robust &= ok
    gt = vol._tsdf.cpu().numpy().astype(np.float64)
    gw = vol._weight.cpu().numpy().astype(np.float64)
    assert robust.mean() > 0.9 and (gw[robust] > 0).mean() > 0.05
    assert np.array_equal(gw[robust], w[robust]), 'the sets of updated voxels differ away from rounding ties'
    assert np.abs(gt[robust] - t[robust]).max() <= 2e-6


def test_fused_volume_feeds_the_renderer():
    """The fused buffer is consumed in place as the permuted [1,1,Z,Y,X] view of get_tsdf.py:95-97."""
    sc = synthetic.mini_scene(device=DEV)
    vol = TSDFVolume(sc.bound.numpy(), 0.04, device=DEV)
    for color, depth, K, pose in frames(sc, 4):
        vol.integrate(color, depth, K, pose)
    tsdf, bnds = vol.get_render_volume()
    assert tsdf.shape[:2] == (1, 1) and tsdf.stride(2) == 1 and not tsdf.is_contiguous()
    import attentive_dfprior_amd as A
    from conftest import make_cfg
    dec = A.DF(); dec.load_state_dict(O.random_state_dict(3)); dec.bound = sc.bound; dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(), None, sc)
    ro, rd, gd, gc = synthetic.make_ray_batch(sc, 64, seed=1)
    with torch.no_grad():
        d, u, col, w = rend.render_batch_ray(sc.c, dec, rd.to(DEV), ro.to(DEV), DEV, tsdf, bnds.to(DEV), 'color', gt_depth=gd.to(DEV))
    od, ou, oc, ow = O.render_batch_ray(O.random_state_dict(3), {k: v.cpu() for k, v in sc.c.items()}, rd, ro, tsdf.cpu(), bnds,
                                        sc.bound, 'color', gd, 32, 16)
    assert ((d.cpu() - od).abs().max() / od.abs().max()).item() < 1e-4
    assert (w != 1).any()


def test_corner_block_copy_follows_integrate():
    """TSDFVolume.integrate writes the volume through a raw pointer.  It bumps the tensor's version, so the cached
    corner-block copy (Engine.tsdf_blocks, what incoherent batches and MapperIteration read) is rebuilt: integrate, render through
    the copy, integrate ANOTHER frame, render again -- both renders equal the plain-volume path bit for bit, the copy was replaced,
    and a writer the version counter does not see is covered by Renderer.invalidate_tsdf()."""
    sc = synthetic.mini_scene(device=DEV)
    vol = TSDFVolume(sc.bound.numpy(), 0.04, device=DEV)
    fr = list(frames(sc, 4))
    for color, depth, K, pose in fr[:2]:
        vol.integrate(color, depth, K, pose)
    tsdf, bnds = vol.get_render_volume()
    import attentive_dfprior_amd as A
    from conftest import make_cfg
    dec = A.DF(); dec.load_state_dict(O.random_state_dict(3)); dec.bound = sc.bound; dec = dec.to(DEV)
    rend = A.Renderer(make_cfg(), None, sc)
    eng = rend._engine
    ro, rd, gd, gc = [t.to(DEV) for t in synthetic.make_ray_batch(sc, 900, seed=2)]
    tb = bnds.to(DEV)

    def render(blocks):
        with torch.no_grad():
            return eng.render_forward(dec, sc.c, ro, rd, gd, tsdf, tb, sc.bound, 'color', 32, 16, tsdf_blocks=blocks)[:4]
    v0 = tsdf._version
    a_plain, a_block = render(False), render(True)
    cb0 = eng._tsdf_cb[1]
    for x, y in zip(a_plain, a_block):
        assert torch.equal(x, y)
    color, depth, K, pose = fr[2]
    vol.integrate(color, depth, K, pose)
    assert tsdf._version > v0, 'integrate must move the version counter the caches are keyed on'
    b_plain, b_block = render(False), render(True)
    assert eng._tsdf_cb[1] is not cb0
    for x, y in zip(b_plain, b_block):
        assert torch.equal(x, y), 'the render through the corner-block copy must see the newly fused frame'
    assert not torch.equal(a_plain[0], b_plain[0]), 'the third frame changed the volume inside the band (the test would be vacuous otherwise)'
    # a writer PyTorch does not see (raw pointer / another process): the hook
    cb1 = eng._tsdf_cb[1]
    w = vol._tsdf.data_ptr()
    half = (vol._tsdf * 0.5).contiguous()
    import ctypes as C
    hip = C.CDLL('libamdhip64.so')
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(w, half.data_ptr(), half.numel() * 4, 3) == 0            # device-to-device, behind PyTorch's back
    torch.cuda.synchronize()
    stale = render(True)
    assert eng._tsdf_cb[1] is cb1                                                  # nobody told the cache: it is stale, by construction
    rend.invalidate_tsdf()
    c_plain, c_block = render(False), render(True)
    assert eng._tsdf_cb[1] is not cb1
    for x, y in zip(c_plain, c_block):
        assert torch.equal(x, y)
    assert not torch.equal(stale[0], c_block[0])
