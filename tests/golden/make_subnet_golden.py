"""
Golden vectors for direct calls of the reference's public sub-modules -- ``decoders.low_decoder(p, c_grid)``,
``decoders.high_decoder``, ``decoders.color_decoder`` (MLP.forward, src/conv_onet/models/decoder.py:177-203) and
``decoders.mlp(p, occ, tsdf_volume, tsdf_bnds)`` (mlp_tsdf.forward, :240-258) -- produced by running the REFERENCE's own
modules (imported read-only through oracle/ref_import.py) on the committed mini scene.  Build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_subnet_golden.py      ->  tests/golden/mini_subnets.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import ref_import                # noqa: E402
from conftest import Mini                    # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    mini = Mini()
    df, rend, rcommon = ref_import.make_reference_objects(mini, mini.sd, mini.n_samples, mini.n_surface)
    qp = mini.query_points                                     # [400,3] f64: inside / band / outside bound / outside the volume
    out = {}
    with torch.no_grad():
        p = qp.unsqueeze(0)
        for name in ('low', 'high', 'color'):
            o = getattr(df, name + '_decoder')(p, mini.c)
            out[name] = o.numpy()
            out[name + '_f32'] = getattr(df, name + '_decoder')(p.float(), mini.c).numpy()     # float32 points (Mesher.py:315)
        g = torch.Generator().manual_seed(21)
        occ = torch.randn(qp.shape[0], generator=g) * 2.0
        fused, w = df.mlp(p, occ, mini.tsdf_volume, mini.tsdf_bnds)
        out.update(att_occ_in=occ.numpy(), att_fused=fused.numpy(), att_w=w.numpy())
    out['source_lines'] = np.array('src/conv_onet/models/decoder.py:177-203, :240-258')
    np.savez_compressed(os.path.join(OUT, 'mini_subnets.npz'), **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == '__main__':
    main()
