"""CPU: the C-ABI library loads, exports every symbol include/adfp.h declares, and the ctypes
structs of attentive_dfprior_amd/_lib.py have the C layout (checked with gcc).  No compute call is
made here (no GPU)."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT
from attentive_dfprior_amd import _lib

HEADER = os.path.join(ROOT, 'include', 'adfp.h')


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(adfp_[a-z_0-9]+)\s*\(', src)))


def test_library_is_built():
    assert os.path.exists(_lib.LIB_PATH), 'run __graft_entry__.build() first'


def test_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 15
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f'{n} declared in include/adfp.h but not exported'
    bound = {s[0] for s in _lib.SYMBOLS}
    assert bound == set(names), (sorted(bound ^ set(names)))


def test_version_and_sizes():
    L = _lib.lib()
    assert L.adfp_version() == _lib.ABI_VERSION
    # parameter counts of the reference's modules (decoder.py:110-166, :212-228)
    assert [L.adfp_decoder_flat_floats(k) for k in range(3)] == [15800, 20920, 15899]
    assert L.adfp_attention_flat_floats() == 33410
    assert L.adfp_decoder_flat_floats(7) < 0
    # every network's packed image must fit the 160 KiB LDS of one CU
    for k in range(3):
        assert L.adfp_decoder_packed_floats(k) * 4 <= 160 * 1024
    assert L.adfp_attention_packed_floats() * 4 <= 160 * 1024
    assert L.adfp_workspace_bytes(0) >= 256
    assert L.adfp_workspace_bytes(1000) >= 1000 * 37


def test_host_side_argument_errors_need_no_gpu():
    L = _lib.lib()
    assert L.adfp_relayout_grid(None, None, 32, 1, 1, 1, None) == -1
    assert L.adfp_pack_decoder(0, None, None, None) == -1
    assert L.adfp_composite(None, None, 1, 1, None, None, None, None, None) == -1
    assert L.adfp_render_forward(None, None, None) == -1


@pytest.mark.skipif(subprocess.call(['which', 'gcc'], stdout=subprocess.DEVNULL) != 0, reason='no gcc')
def test_ctypes_struct_layout_matches_c(tmp_path):
    prog = tmp_path / 'layout.c'
    prog.write_text('''
#include <stdio.h>
#include <stddef.h>
#include "adfp.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(adfp_grid), sizeof(adfp_tsdf), sizeof(adfp_scene), sizeof(adfp_points),
         sizeof(adfp_render_args), sizeof(adfp_train_state), sizeof(adfp_backward_args), sizeof(adfp_loss_args));
  printf("%zu %zu %zu %zu %zu\\n", offsetof(adfp_scene, low), offsetof(adfp_scene, tsdf), offsetof(adfp_scene, w_low),
         offsetof(adfp_scene, w_att), offsetof(adfp_scene, status));
  printf("%zu %zu %zu\\n", offsetof(adfp_points, pts), offsetof(adfp_points, z_vals), offsetof(adfp_points, S));
  printf("%zu %zu %zu %zu %zu\\n", offsetof(adfp_render_args, perturb), offsetof(adfp_render_args, rays_o),
         offsetof(adfp_render_args, workspace), offsetof(adfp_render_args, workspace_bytes), offsetof(adfp_render_args, state));
  printf("%zu %zu %zu %zu %zu\\n", offsetof(adfp_backward_args, rays_o), offsetof(adfp_backward_args, state),
         offsetof(adfp_backward_args, g_depth), offsetof(adfp_backward_args, g_rays_d), offsetof(adfp_backward_args, workspace_bytes));
  printf("%zu %zu %zu %zu\\n", offsetof(adfp_backward_args, ray_keep), offsetof(adfp_loss_args, w_color_loss), offsetof(adfp_loss_args, depth),
         offsetof(adfp_loss_args, g_weight));
  printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(adfp_adam_group), offsetof(adfp_adam_group, mask), offsetof(adfp_adam_group, channels),
         offsetof(adfp_adam_group, derived), offsetof(adfp_scene, ht_low), offsetof(adfp_train_state, masks_low),
         offsetof(adfp_train_state, act_color));
  printf("%zu %zu %zu %zu %zu\\n", sizeof(adfp_frame_job), offsetof(adfp_frame_job, fx), offsetof(adfp_frame_job, depth),
         offsetof(adfp_frame_job, rays_d), offsetof(adfp_render_args, frame));
  return 0;
}''')
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-std=c99', '-I', os.path.join(ROOT, 'include'), str(prog), '-o', str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split('\n')
    sz = list(map(int, out[0].split()))
    assert sz == [ctypes.sizeof(_lib.AdfpGrid), ctypes.sizeof(_lib.AdfpTsdf), ctypes.sizeof(_lib.AdfpScene),
                  ctypes.sizeof(_lib.AdfpPoints), ctypes.sizeof(_lib.AdfpRenderArgs),
                  ctypes.sizeof(_lib.AdfpTrainState), ctypes.sizeof(_lib.AdfpBackwardArgs), ctypes.sizeof(_lib.AdfpLossArgs)]
    S = _lib.AdfpScene
    assert list(map(int, out[1].split())) == [S.low.offset, S.tsdf.offset, S.w_low.offset, S.w_att.offset, S.status.offset]
    P = _lib.AdfpPoints
    assert list(map(int, out[2].split())) == [P.pts.offset, P.z_vals.offset, P.S.offset]
    R = _lib.AdfpRenderArgs
    assert list(map(int, out[3].split())) == [R.perturb.offset, R.rays_o.offset, R.workspace.offset,
                                              R.workspace_bytes.offset, R.state.offset]
    B = _lib.AdfpBackwardArgs
    assert list(map(int, out[4].split())) == [B.rays_o.offset, B.state.offset, B.g_depth.offset, B.g_rays_d.offset,
                                              B.workspace_bytes.offset]
    Lo = _lib.AdfpLossArgs
    assert list(map(int, out[5].split())) == [B.ray_keep.offset, Lo.w_color_loss.offset, Lo.depth.offset, Lo.g_weight.offset]
    G, T = _lib.AdfpAdamGroup, _lib.AdfpTrainState
    assert list(map(int, out[6].split())) == [ctypes.sizeof(G), G.mask.offset, G.channels.offset, G.derived.offset, S.ht_low.offset,
                                              T.masks_low.offset, T.act_color.offset]
    F = _lib.AdfpFrameJob
    assert list(map(int, out[7].split())) == [ctypes.sizeof(F), F.fx.offset, F.depth.offset, F.rays_d.offset, R.frame.offset]


def test_reference_fusion_kernel_builds_as_a_checker():
    """oracle/_ref: the reference's CUDA C kernel string (src/fusion.py:69-142) compiled by hipcc from where it lies, in the
    as-written and the contracted variant; both export the launcher tests/test_gpu_fusion.py calls on the GPU box."""
    import ctypes
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists('/root/reference/src/fusion.py'):
        pytest.skip('/root/reference is not present (GPU box): the prebuilt oracle/_ref files are used as they are')
    sys.path.insert(0, os.path.join(root, 'oracle'))
    try:
        import build_ref_fusion
        assert 'SourceModule' not in build_ref_fusion.kernel_string() and '__global__ void integrate' in build_ref_fusion.kernel_string()
        assert build_ref_fusion.build()
    finally:
        sys.path.pop(0)
    for name in ('libref_fusion_exact.so', 'libref_fusion_contract.so'):
        lib = ctypes.CDLL(os.path.join(root, 'oracle', '_ref', name))
        assert hasattr(lib, 'ref_fusion_integrate')
    # nothing of the reference's text is kept in the repository
    for dirpath, _, files in os.walk(os.path.join(root, 'oracle')):
        for f in files:
            if f.endswith(('.py', '.hip', '.cu', '.txt')):
                assert 'float voxel_x = floorf' not in open(os.path.join(dirpath, f), errors='ignore').read(), f
