"""
ORACLE SUPPORT -- TEST INFRASTRUCTURE ONLY, BUILD CONTAINER ONLY.

Imports the reference's hot-path modules read-only from /root/reference so that the oracle
(oracle/adfp_oracle.py) can be validated against the real thing and golden fixtures can be
generated (tests/golden/make_golden.py).  /root/reference does not exist on the GPU box and
nothing under tests -m gpu / smoke / bench may import this module.

Shim: the reference hard-codes CUDA device strings (src/conv_onet/models/decoder.py:241
'cuda:0', :312 f'cuda:{p.get_device()}' == 'cuda:-1' on CPU).  We map any 'cuda*' device
string to 'cpu' inside torch.Tensor.to for the duration of the import/use.  The reference's
files are not modified (PYTHONDONTWRITEBYTECODE keeps __pycache__ untouched).
"""
import os
import sys
import types

import torch

REF = os.environ.get('ADFP_REFERENCE', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REF, 'src'))


_orig_to = torch.Tensor.to


def _to_cpu_shim(self, *args, **kwargs):
    args = list(args)
    for i, a in enumerate(args):
        if isinstance(a, str) and a.startswith('cuda'):
            args[i] = 'cpu'
        elif isinstance(a, int) and not isinstance(a, bool) and a < 0:      # .to(t.get_device()) of a CPU tensor (src/common.py:152)
            args[i] = 'cpu'
    if isinstance(kwargs.get('device'), str) and kwargs['device'].startswith('cuda'):
        kwargs['device'] = 'cpu'
    return _orig_to(self, *args, **kwargs)


def load():
    """Returns (ref_common, ref_decoder, ref_Renderer_module)."""
    if not available():
        raise RuntimeError(f'reference checkout not found at {REF}')
    sys.dont_write_bytecode = True
    torch.Tensor.to = _to_cpu_shim
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import importlib
    common = importlib.import_module('src.common')
    decoder = importlib.import_module('src.conv_onet.models.decoder')
    renderer = importlib.import_module('src.utils.Renderer')
    return common, decoder, renderer


def make_reference_objects(scene, sd, n_samples, n_surface, lindisp=False, perturb=0.0):
    """Reference DF module loaded with state dict `sd` + reference Renderer bound to `scene`."""
    common, decoder, renderer = load()
    df = decoder.DF(dim=3, c_dim=32, low_grid_len=0.32, high_grid_len=0.16, color_grid_len=0.16,
                    hidden_size=32, pos_embedding_method='fourier')
    df.load_state_dict(sd)
    df.bound = scene.bound
    for m in (df.low_decoder, df.high_decoder, df.color_decoder):
        m.bound = scene.bound                                   # src/DF_Prior.py:191-194
    cfg = {'rendering': {'lindisp': lindisp, 'perturb': perturb, 'N_samples': n_samples,
                         'N_surface': n_surface, 'N_importance': 0},
           'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}
    slam = types.SimpleNamespace(bound=scene.bound, vol_bnds=scene.tsdf_bnds, H=scene.H, W=scene.W,
                                 fx=scene.fx, fy=scene.fy, cx=scene.cx, cy=scene.cy)
    rend = renderer.Renderer(cfg, None, slam)
    return df, rend, common
