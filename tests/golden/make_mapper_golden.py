"""
Golden vectors for the two Mapper-side rows whose module (src/Mapper.py) cannot be IMPORTED in the build container
(it needs cv2 / colorama): the script reads the reference's own source lines from /root/reference and EXECUTES them on
seeded inputs, so the stored outputs come from the reference's code, not from a restatement.  Build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_mapper_golden.py

  mapper_prefilter.npz   a3, src/Mapper.py:438-449 (the ten inline lines of optimize_map that drop rays whose sensor
                         depth lies outside the bounding box), executed verbatim with `self.bound`, `device` and the
                         four batch tensors bound in the namespace; cases with zero direction components (+-inf),
                         0/0 planes (NaN), non-finite rays and depth == t ties.
  mapper_frustum.npz     f4, src/Mapper.py:90-158 (Mapper.get_mask_from_c2w) executed verbatim as a method of a stub
                         `self`, with ONE substitution: the name `cv2` resolves to a namespace whose `remap` is the
                         oracle's restatement of OpenCV 4.5.5's bilinear remap (opencv-python==4.5.5.64,
                         environment.yaml:194; absent here, no network) -- everything around the remap call is the
                         reference's own numpy / torch code.
"""
import os
import sys
import textwrap
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import adfp_oracle as O          # noqa: E402
from attentive_dfprior_amd import synthetic  # noqa: E402

REF = os.environ.get('ADFP_REFERENCE', '/root/reference')
OUT = os.path.dirname(os.path.abspath(__file__))


def ref_lines(first, last, must_start, must_end):
    src = open(os.path.join(REF, 'src', 'Mapper.py')).read().split('\n')
    block = src[first - 1:last]
    assert must_start in block[0], (block[0], must_start)
    assert must_end in block[-1], (block[-1], must_end)
    return textwrap.dedent('\n'.join(block))


def prefilter_cases():
    """(name, bound f64 [3,2], rays_o, rays_d, depth, color)"""
    cases = []
    sc = synthetic.mini_scene()
    g = torch.Generator().manual_seed(17)
    for n in (1, 63, 1024, 1025, 5000):
        ro, rd, depth, color = synthetic.make_ray_batch(sc, n, seed=40 + n, zero_frac=0.2)
        depth = depth * (0.5 + 1.5 * torch.rand(depth.shape, generator=g))      # some beyond the box
        if n >= 63:
            rd[3, 0] = 0.0                                   # +-inf on one axis
            rd[5] = 0.0                                      # inf / NaN everywhere
            rd[11, 2] = float('nan')
            ro[12, 1] = float('inf')
            t = (sc.bound.unsqueeze(0) - ro[9:10].unsqueeze(-1)) / rd[9:10].unsqueeze(-1)
            depth[9] = torch.min(torch.max(t, dim=2)[0], dim=1)[0].float()      # t == depth up to the f32 rounding
        cases.append((f'mini{n}', sc.bound, ro, rd, depth, color))
    # a bound made of binary fractions: an f32 origin can sit EXACTLY on an f64 bound plane -> 0/0 = NaN
    bound = torch.tensor([[-1.0, 1.5], [-1.0, 1.25], [-0.75, 1.0]], dtype=torch.float64)
    ro = torch.tensor([[0.25, 0.125, 0.125], [-1.0, 0.125, 0.125], [1.5, 0.125, 0.125], [0.25, 1.25, 0.125],
                       [0.25, 0.125, 0.125], [0.25, 0.125, 0.125], [-1.0, 0.125, 0.125], [0.25, 0.125, 1.0]])
    rd = torch.tensor([[0.0, 0.3, -1.0], [0.0, 0.3, -1.0], [0.0, -0.2, -1.0], [0.2, 0.0, -1.0],
                       [0.0, 0.0, -1.0], [0.1, 0.2, -1.0], [0.5, 0.1, -1.0], [0.1, 0.1, 0.0]])
    depth = torch.tensor([0.3, 0.3, 0.0, 0.25, 0.875, 0.3, 0.4, 0.2])
    cases.append(('exact', bound, ro, rd, depth, torch.rand(8, 3, generator=g)))
    return cases


def make_prefilter():
    code = ref_lines(438, 449, '# should pre-filter those out of bounding box depth value', 'batch_gt_color = batch_gt_color[inside_mask]')
    out = {'source_lines': np.array('src/Mapper.py:438-449')}
    for name, bound, ro, rd, depth, color in prefilter_cases():
        ns = {'torch': torch, 'self': types.SimpleNamespace(bound=bound), 'device': 'cpu',
              'batch_rays_o': ro.clone(), 'batch_rays_d': rd.clone(), 'batch_gt_depth': depth.clone(), 'batch_gt_color': color.clone()}
        exec(code, ns)
        out[f'{name}.bound'] = bound.numpy()
        out[f'{name}.rays_o'], out[f'{name}.rays_d'] = ro.numpy(), rd.numpy()
        out[f'{name}.gt_depth'], out[f'{name}.gt_color'] = depth.numpy(), color.numpy()
        out[f'{name}.inside_mask'] = ns['inside_mask'].numpy()
        out[f'{name}.kept_rays_o'] = ns['batch_rays_o'].numpy()
        out[f'{name}.kept_gt_depth'] = ns['batch_gt_depth'].numpy()
        print(name, 'kept', int(ns['inside_mask'].sum()), 'of', ro.shape[0])
    np.savez_compressed(os.path.join(OUT, 'mapper_prefilter.npz'), **out)


def make_frustum():
    code = ref_lines(90, 158, 'def get_mask_from_c2w(self, c2w, key, val_shape, depth_np):', 'return mask')
    cv2_stub = types.SimpleNamespace(INTER_LINEAR=1,
                                     remap=lambda img, mx, my, interpolation: O.remap_linear_np(img, mx, my).reshape(-1, 1))
    ns = {'torch': torch, 'np': np, 'cv2': cv2_stub}
    exec(code, ns)
    fn = ns['get_mask_from_c2w']
    out = {'source_lines': np.array('src/Mapper.py:90-158')}
    sc = synthetic.mini_scene()
    for k, (yaw, pitch, off) in enumerate([(0.7, 0.15, (0.05, -0.03, 0.02)), (2.9, -0.3, (-0.1, 0.1, 0.0)), (4.4, 0.0, (0.2, 0.0, -0.1))]):
        c2w = sc.default_c2w(offset=off, yaw=yaw, pitch=pitch)
        depth = sc.depth_image(c2w, zero_band=0.1).numpy()
        stub = types.SimpleNamespace(H=sc.H, W=sc.W, fx=sc.fx, fy=sc.fy, cx=sc.cx, cy=sc.cy, bound=sc.bound)
        out[f'pose{k}.c2w'], out[f'pose{k}.depth'] = c2w.numpy(), depth
        for key, val in sc.c.items():
            mask = fn(stub, c2w, key, val.shape[2:], depth)           # [X, Y, Z] bool, as the reference returns it
            out[f'pose{k}.{key}'] = np.ascontiguousarray(mask)
            print(k, key, mask.shape, int(mask.sum()), 'of', mask.size)
    out['bound'] = sc.bound.numpy()
    out['intrinsics'] = np.array([sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'mapper_frustum.npz'), **out)


if __name__ == '__main__':
    make_prefilter()
    make_frustum()
