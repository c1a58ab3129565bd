"""
Golden vectors for the Tracker's pose utilities -- ``quad2rotation`` / ``get_camera_from_tensor`` (src/common.py:139-178) and
their autograd -- produced by running the REFERENCE's own functions (imported read-only through oracle/ref_import.py).  Build
container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_pose_golden.py      ->  tests/golden/mini_pose.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_import                # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    rcommon, _, _ = ref_import.load()
    g = torch.Generator().manual_seed(77)
    n = 24
    cam = torch.randn(n, 7, generator=g)
    cam[::2, :4] /= cam[::2, :4].norm(dim=1, keepdim=True)          # every other one a unit quaternion, what the Tracker starts from
    cot = torch.randn(n, 3, 4, generator=g)                          # a cotangent of the [3,4] camera matrix
    rts, grads = [], []
    for k in range(n):
        t = cam[k].clone().requires_grad_(True)
        RT = rcommon.get_camera_from_tensor(t)                       # [3,4]
        (RT * cot[k]).sum().backward()
        rts.append(RT.detach().numpy())
        grads.append(t.grad.numpy())
    batch = rcommon.get_camera_from_tensor(cam)                      # the batched form
    np.savez_compressed(os.path.join(OUT, 'mini_pose.npz'), cam=cam.numpy(), cot=cot.numpy(), c2w=np.stack(rts), g_cam=np.stack(grads),
                        c2w_batched=batch.numpy(), source_lines=np.array('src/common.py:139-178'))
    print('mini_pose.npz', n, 'poses')


if __name__ == '__main__':
    main()
