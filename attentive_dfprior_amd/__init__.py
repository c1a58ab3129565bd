"""attentive_dfprior_amd -- MI355X-native per-ray volume-rendering hot path of Attentive_DFPrior.

Public surface mirrors the reference's modules for this path:
  Renderer  <-> src/utils/Renderer.py        DF/MLP/mlp_tsdf <-> src/conv_onet/models/decoder.py
  common    <-> src/common.py (rays, compositing)
All arithmetic runs in libadfp.so (hand-written HIP for gfx950); see include/adfp.h.
"""
from .decoder import DF, MLP, mlp_tsdf, DenseLayer, GaussianFourierFeatureTransform  # noqa: F401
from .renderer import Renderer  # noqa: F401
from . import common  # noqa: F401

decoder_dict = {'dfprior': DF}   # src/conv_onet/models/__init__.py:4


def get_model(cfg):
    """Decoder factory with the reference's config keys (src/conv_onet/config.py:4-27)."""
    return decoder_dict['dfprior'](
        dim=cfg['data']['dim'], c_dim=cfg['model']['c_dim'],
        low_grid_len=cfg['grid_len']['low'], high_grid_len=cfg['grid_len']['high'],
        color_grid_len=cfg['grid_len']['color'],
        pos_embedding_method=cfg['model']['pos_embedding_method'])
