"""CPU: host-side mirror of the reference interface (module tree, parameter names, config
handling, CPU ray helpers, no-fallback behaviour)."""
import copy
import pickle

import numpy as np
import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import common, synthetic
from oracle import adfp_oracle as O
from conftest import make_cfg


def test_state_dict_names_and_shapes_match_reference():
    m = A.DF()
    want = O.decoder_param_shapes()
    got = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert got == [(k, tuple(s)) for k, s in want]
    assert sum(p.numel() for p in m.parameters()) == 86029


def test_load_state_dict_deepcopy_pickle(mini):
    m = A.DF()
    m.load_state_dict(mini.sd)
    m.bound = mini.bound
    m2 = copy.deepcopy(m)                       # src/Tracker.py:144
    assert m2._packed == {} and torch.equal(m2.mlp.output_linear.weight, m.mlp.output_linear.weight)
    m3 = pickle.loads(pickle.dumps(m))          # mp.spawn, src/DF_Prior.py:302-311
    assert torch.equal(m3.low_decoder.embedder._B, m.low_decoder.embedder._B)
    for name in ('low_decoder', 'high_decoder', 'color_decoder', 'mlp'):
        assert len(list(getattr(m, name).parameters())) > 0   # src/Mapper.py:364-371
    m.share_memory()                            # src/DF_Prior.py:108-110


def test_get_model_uses_reference_config_keys():
    cfg = {'data': {'dim': 3}, 'grid_len': {'low': 0.32, 'high': 0.16, 'color': 0.16},
           'model': {'c_dim': 32, 'pos_embedding_method': 'fourier'}}
    m = A.get_model(cfg)
    assert isinstance(m, A.DF) and m.high_decoder.c_dim == 64 and m.color_decoder.color
    cfg['model']['pos_embedding_method'] = 'nerf'
    with pytest.raises(NotImplementedError):
        A.get_model(cfg)


def test_renderer_reads_cfg_like_reference(mini):
    r = A.Renderer(make_cfg(48, 16), None, mini)
    assert (r.N_samples, r.N_surface, r.ray_batch_size, r.points_batch_size) == (48, 16, 100000, 500000)
    assert r.bound is mini.bound and r.tsdf_bnds is mini.vol_bnds
    cfg = make_cfg()
    cfg['rendering']['N_importance'] = 8
    with pytest.raises(NotImplementedError):
        A.Renderer(cfg, None, mini)
    cfg = make_cfg()
    cfg['occupancy'] = False
    with pytest.raises(NotImplementedError):
        A.Renderer(cfg, None, mini)


def test_no_cpu_fallback(mini):
    """The product path must fail loudly on CPU tensors instead of computing somewhere else."""
    m = A.DF()
    m.load_state_dict(mini.sd)
    m.bound = mini.bound
    r = A.Renderer(make_cfg(), None, mini)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        r.render_batch_ray(mini.c, m, mini.rays_d, mini.rays_o, 'cpu', mini.tsdf_volume, mini.tsdf_bnds, 'color',
                           mini.gt_depth)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(mini.query_points.unsqueeze(0), c_grid=mini.c, tsdf_volume=mini.tsdf_volume, tsdf_bnds=mini.tsdf_bnds,
          stage='high')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        common.raw2outputs_nerf_color(torch.zeros(2, 4, 4), torch.zeros(2, 4), None, occupancy=True)


def test_cpu_ray_helpers_match_golden(mini):
    g = mini.golden('rays')
    ro, rd = common.get_rays(mini.H, mini.W, mini.fx, mini.fy, mini.cx, mini.cy, mini.c2w, 'cpu')
    assert np.abs(rd.numpy() - g['get_rays_d']).max() <= 1e-6
    assert np.array_equal(ro.contiguous().numpy(), g['get_rays_o'])
    ro2, rd2 = common.get_rays_from_uv(torch.from_numpy(g['uv_i']), torch.from_numpy(g['uv_j']), mini.c2w,
                                       mini.H, mini.W, mini.fx, mini.fy, mini.cx, mini.cy, 'cpu')
    assert np.abs(rd2.numpy() - g['uv_rays_d']).max() <= 1e-6


def test_get_samples_uses_torch_randint_stream(mini):
    depth = mini.depth_img
    color = torch.rand(mini.H, mini.W, 3)
    torch.manual_seed(4)
    ro, rd, d, c = common.get_samples(0, mini.H, 0, mini.W, 50, mini.H, mini.W, mini.fx, mini.fy, mini.cx, mini.cy,
                                      mini.c2w, depth, color, 'cpu')
    torch.manual_seed(4)
    idx = torch.randint(mini.H * mini.W, (50,))
    assert torch.equal(d, depth.reshape(-1)[idx]) and torch.equal(c, color.reshape(-1, 3)[idx])
    i, j, d2, c2 = O.select_uv(0, mini.H, 0, mini.W, idx, depth, color)
    _, rd_ref = O.get_rays_from_uv(i, j, mini.c2w, mini.fx, mini.fy, mini.cx, mini.cy)
    assert (rd - rd_ref).abs().max().item() <= 1e-6


def test_synthetic_scene_shapes_follow_reference_formulas():
    b = synthetic.scene_bound(synthetic.SCENE_BOUNDS['room0'])
    assert torch.allclose(b[:, 1], torch.tensor([8.94, 5.76, 3.54], dtype=torch.float64), atol=1e-9)
    assert synthetic.grid_shape(b, 0.16)[2:] == [43, 56, 74]
    assert synthetic.grid_shape(b, 0.32)[2:] == [21, 28, 37]
    sc = synthetic.mini_scene()
    assert sc.tsdf_volume.shape[:2] == (1, 1) and not sc.tsdf_volume.is_contiguous()
    assert sc.tsdf_volume.stride(2) == 1                           # Z fastest, as get_tsdf.py:95-97
    assert float(sc.tsdf_volume.min()) == -1.0 and float(sc.tsdf_volume.max()) == 1.0


def test_get_tensor_from_camera_normalises_like_mathutils():
    """The reference goes through mathutils' Matrix.to_quaternion (src/common.py:181-203), which normalises the matrix and returns a
    UNIT quaternion.  The Tracker's constant-speed guess delta @ pre_c2w (src/Tracker.py:213-215) is orthonormal only up to float
    error: the camera tensor made from it must still be a unit quaternion (Adam steps on the raw components), and the rotation
    of a uniformly or per-axis scaled matrix is the rotation of the unscaled one."""
    import torch
    from attentive_dfprior_amd import common
    g = torch.Generator().manual_seed(0)
    for k in range(8):
        cam = torch.randn(7, generator=g)
        RT = common.get_camera_from_tensor(cam)
        ref = common.get_tensor_from_camera(RT)
        skew = RT.clone()
        skew[:3, :3] *= torch.tensor([1.0 + 2e-3, 1.0 - 1e-3, 1.0 + 5e-4])          # per-column scale: unit axis vectors restore R
        got = common.get_tensor_from_camera(skew)
        assert abs(float(got[:4].norm()) - 1.0) < 1e-6
        assert (got - ref).abs().max() < 1e-5, (got, ref)
        noisy = RT.clone()
        noisy[:3, :3] += 1e-4 * torch.randn(3, 3, generator=g)                     # accumulated float error: not exactly a rotation
        q = common.get_tensor_from_camera(noisy)
        assert abs(float(q[:4].norm()) - 1.0) < 1e-6 and (q - ref).abs().max() < 1e-3


def test_get_tensor_from_camera_against_scipy_rotation():
    """mathutils (what src/common.py:192-195 converts the matrix with) is absent from the image, so the conversion has no vector of
    the reference's own.  An INDEPENDENT implementation of the same map pins it instead: scipy.spatial.transform.Rotation
    (matrix -> unit quaternion), over random rotations, rotations by almost 180 degrees about every axis (the trace <= 0
    branches) and the identity -- the same rotation, with the sign the r >= 0 convention picks."""
    import numpy as np
    import torch
    from scipy.spatial.transform import Rotation
    from attentive_dfprior_amd import common
    rng = np.random.default_rng(5)
    mats = [Rotation.random(random_state=int(k)).as_matrix() for k in range(200)]
    for axis in np.eye(3):
        for ang in (np.pi - 1e-3, np.pi - 1e-6, np.pi, 2.5, -3.0):
            mats.append(Rotation.from_rotvec(axis * ang).as_matrix())
    for k in range(40):                                                   # near-180-degree turns about random axes
        v = rng.normal(size=3)
        mats.append(Rotation.from_rotvec(v / np.linalg.norm(v) * (np.pi - 10.0 ** -rng.uniform(1, 7))).as_matrix())
    mats.append(np.eye(3))
    for R in mats:
        RT = np.eye(4)
        RT[:3, :3] = R
        RT[:3, 3] = rng.normal(size=3)
        got = common.get_tensor_from_camera(torch.from_numpy(RT)).double().numpy()
        x, y, z, w = Rotation.from_matrix(R).as_quat()
        ref = np.array([w, x, y, z])
        ref = -ref if ref[0] < 0 else ref
        q = got[:4]
        assert abs(np.linalg.norm(q) - 1.0) < 1e-6
        assert min(np.abs(q - ref).max(), np.abs(q + ref).max()) < 2e-6, (q, ref)      # (r = 0 exactly: either sign is the r >= 0 branch)
        assert q[0] >= 0
        assert np.abs(got[4:] - RT[:3, 3]).max() < 1e-6
        # and it is the quaternion get_camera_from_tensor (src/common.py:165-178, pinned by mini_pose.npz) turns back into R
        back = common.get_camera_from_tensor(torch.from_numpy(got).float()).double().numpy()
        assert np.abs(back[:3, :3] - R).max() < 5e-6



def test_design_md_is_the_template_filled_from_profiles():
    """DESIGN.md quotes profiles/ and nothing else: it is tools/DESIGN.template.md with its @@fields@@ read out of the committed
    profile files by tools/design_numbers.py.  To change the text edit the TEMPLATE; after a new collection regenerate:
        python tools/design_numbers.py tools/DESIGN.template.md profiles > DESIGN.md"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'design_numbers.py'), os.path.join(root, 'tools', 'DESIGN.template.md'),
                        os.path.join(root, 'profiles')], capture_output=True, text=True, cwd=root)
    assert r.returncode == 0, r.stderr[-400:]
    assert 'unfilled' not in r.stderr, r.stderr[-400:]
    assert r.stdout == open(os.path.join(root, 'DESIGN.md')).read(), \
        'DESIGN.md differs from the filled template: edit tools/DESIGN.template.md and regenerate (see this test\'s docstring)'


def test_limiter_model_input_is_this_build():
    """bench.py's `roofline.limiter` prices one tile of k_decode_lc16 with the instruction counts of its TILE LOOP, read from
    profiles/r06_isa_mix_lc16.txt.  The file is generated from the sources (python tools/gen_isa_mix.py: hipcc -S + tools/isa_mix.py
    --loop) and stamped with their hash: a kernel change without a regeneration fails here instead of citing a stale count
    (the model quoted a round-4 whole-kernel count, 2 415 VALU, for a loop that has 2 175)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    mix = bench.isa_mix_lc16()
    assert mix['file'] == 'profiles/r06_isa_mix_lc16.txt'
    assert mix['profiled_source_hash'] == bench.source_hash() and mix['stale'] is False, \
        'the kernel sources changed since profiles/r06_isa_mix_lc16.txt was generated: python tools/gen_isa_mix.py'
    assert mix['mfma'] == 360 and 1500 < mix['valu'] < 2600 and mix['lds'] > 100      # one 32-point tile, both networks: 2 x 180 MFMAs
    text = open(os.path.join(root, 'profiles', 'r06_isa_mix_lc16.txt')).read()
    assert text.index('[tile loop only]') < text.index('12DecodeLCArgs:')               # the loop first, the whole kernel after it
