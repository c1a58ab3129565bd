"""GPU: the C ABI is enough on its own -- examples/abi_demo.cpp (C++ + HIP runtime + include/adfp.h, no Python,
no torch) renders the committed mini scene from raw array dumps and must reproduce the reference's golden
outputs like the Python binding does."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _flat(sd, prefix):
    return np.concatenate([v.numpy().reshape(-1).astype(np.float32) for k, v in sd.items() if k.startswith(prefix)])


def test_cpp_consumer_matches_golden(mini, tmp_path):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc on this box')
    exe = tmp_path / 'abi_demo'
    libdir = os.path.join(ROOT, 'attentive_dfprior_amd')
    subprocess.run([hipcc, '-O2', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'),
                    os.path.join(ROOT, 'examples', 'abi_demo.cpp'), '-L' + libdir, '-ladfp', '-o', str(exe)],
                   check=True, capture_output=True)
    d = tmp_path / 'scene'
    d.mkdir()
    stage = 2                                                        # ADFP_STAGE_COLOR
    grids = [mini.c[k] for k in ('grid_low', 'grid_high', 'grid_color')]
    tsdf_xyz = mini.tsdf_volume[0, 0].permute(2, 1, 0).contiguous()  # back to the physical [X,Y,Z] buffer
    meta = [mini.rays_o.shape[0], mini.n_samples, mini.n_surface, stage]
    for g in grids:
        meta += list(g.shape[2:])
    meta += list(tsdf_xyz.shape)
    np.array(meta, dtype=np.int64).tofile(d / 'meta.i64')
    np.concatenate([mini.bound.numpy().reshape(-1), mini.tsdf_bnds.numpy().reshape(-1)]).astype(np.float64).tofile(d / 'bounds.f64')
    for name, g in zip(('grid_low', 'grid_high', 'grid_color'), grids):
        g[0].numpy().astype(np.float32).tofile(d / f'{name}.f32')
    tsdf_xyz.numpy().astype(np.float32).tofile(d / 'tsdf_xyz.f32')
    for net, prefix in (('low', 'low_decoder.'), ('high', 'high_decoder.'), ('color', 'color_decoder.'), ('att', 'mlp.')):
        _flat(mini.sd, prefix).tofile(d / f'flat_{net}.f32')
    mini.rays_o.numpy().astype(np.float32).tofile(d / 'rays_o.f32')
    mini.rays_d.numpy().astype(np.float32).tofile(d / 'rays_d.f32')
    mini.gt_depth.numpy().astype(np.float32).tofile(d / 'gt_depth.f32')
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + os.pathsep + os.environ.get('LD_LIBRARY_PATH', ''))
    r = subprocess.run([str(exe), str(d)], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    g = mini.golden('color')
    N, S = mini.rays_o.shape[0], mini.n_samples + mini.n_surface
    depth = np.fromfile(d / 'out_depth.f64', dtype=np.float64)
    unc = np.fromfile(d / 'out_uncertainty.f64', dtype=np.float64)
    color = np.fromfile(d / 'out_color.f32', dtype=np.float32).reshape(N, 3)
    weight = np.fromfile(d / 'out_weight.f32', dtype=np.float32).reshape(N, S, 1)
    for got, ref, what in ((depth, g['depth'], 'depth'), (unc, g['uncertainty'], 'uncertainty'), (color, g['color'], 'color'),
                           (weight, g['weight'], 'weight')):
        scale = np.abs(ref).max()
        assert np.all(np.abs(got - ref) <= 1e-4 * (np.abs(ref) + 0.1 * scale)), what
