"""
Drop-in for the integration part of the reference's ``src/fusion.py`` ``TSDFVolume``: same constructor
arguments, ``integrate(color_im, depth_im, cam_intr, cam_pose, obs_weight)`` and ``get_volume()``, with the
volumes resident on the MI355X and the per-voxel update in libadfp.so (``adfp_tsdf_integrate``) instead of a
PyCUDA kernel / numba loops.  ``get_render_volume()`` hands the render path exactly what
``get_tsdf.py:95-97`` builds (the permuted, non-contiguous ``[1,1,Z,Y,X]`` view) without leaving the device.
Mesh extraction (marching cubes, ``get_mesh``) stays the reference's scikit-image code.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import lib, ptr, check


class TSDFVolume(object):
    def __init__(self, vol_bnds, voxel_size, device='cuda:0'):
        # Volume geometry as src/fusion.py:32-44 defines it: the voxel count per axis is the extent divided by the voxel edge, rounded
        # up; the upper bounds then move out to the last voxel's far face; the origin is the lower corner in float32 (it is a kernel
        # argument).  Attribute names are the reference's (get_tsdf.py and the mesh code read them).
        bnds = np.array(vol_bnds, dtype=np.float64)
        if bnds.shape != (3, 2):
            raise AssertionError('[!] `vol_bnds` should be of shape (3, 2).')
        edge = float(voxel_size)
        lower = bnds[:, 0]
        counts = np.ceil((bnds[:, 1] - lower) / edge).astype(int)
        bnds[:, 1] = lower + counts * edge
        self._vol_bnds = bnds
        self._voxel_size = edge
        self._trunc_margin = 5 * edge                                   # src/fusion.py:38
        self._color_const = 256 * 256
        self._vol_dim = np.ascontiguousarray(counts)
        self._vol_origin = np.ascontiguousarray(lower, dtype=np.float32)
        self.device = torch.device(device)
        dims = tuple(int(v) for v in self._vol_dim)
        self._tsdf = torch.full(dims, -1.0, dtype=torch.float32, device=self.device)      # unobserved = -1 (:52)
        self._weight = torch.zeros(dims, dtype=torch.float32, device=self.device)
        self._color = torch.zeros(dims, dtype=torch.float32, device=self.device)

    def integrate(self, color_im, depth_im, cam_intr, cam_pose, obs_weight=1.):
        """Integrate one RGB-D frame (color_im [H,W,3] 0..255, depth_im [H,W], 3x3 intrinsics, 4x4 pose)."""
        color_im = torch.as_tensor(np.asarray(color_im)) if not torch.is_tensor(color_im) else color_im
        depth_im = torch.as_tensor(np.asarray(depth_im)) if not torch.is_tensor(depth_im) else depth_im
        color = color_im.to(self.device, torch.float32)
        packed = torch.floor(color[..., 2] * self._color_const + color[..., 1] * 256 + color[..., 0]).contiguous()
        depth = depth_im.to(self.device, torch.float32).contiguous()
        im_h, im_w = depth.shape
        org = (C.c_float * 3)(*[float(v) for v in self._vol_origin])
        intr = (C.c_float * 9)(*[float(v) for v in np.asarray(cam_intr, dtype=np.float32).reshape(-1)])
        pose = (C.c_float * 16)(*[float(v) for v in np.asarray(cam_pose, dtype=np.float32).reshape(-1)])
        with torch.cuda.device(self.device):
            check(lib().adfp_tsdf_integrate(ptr(self._tsdf), ptr(self._weight), ptr(self._color),
                                            int(self._vol_dim[0]), int(self._vol_dim[1]), int(self._vol_dim[2]),
                                            C.byref(org), np.float32(self._voxel_size).item(), C.byref(intr), C.byref(pose),
                                            ptr(packed), ptr(depth), im_h, im_w, np.float32(self._trunc_margin).item(),
                                            float(obs_weight), _lib.current_stream(self.device)), 'adfp_tsdf_integrate')
        # The kernel wrote the volume through a raw pointer: tell PyTorch.  Every cached conversion of the render volume (the
        # corner-block copy of Engine.tsdf_blocks / MapperIteration) is keyed on (data_ptr, _version), and the view handed out by
        # get_render_volume() shares this counter.
        torch.autograd.graph.increment_version(self._tsdf)

    def get_volume(self):
        """(tsdf [X,Y,Z], color [X,Y,Z], bounds [3,2]) as numpy, like src/fusion.py:297-301."""
        return self._tsdf.cpu().numpy(), self._color.cpu().numpy(), self._vol_bnds

    def get_render_volume(self):
        """The TSDF exactly as the renderer consumes it: the [1,1,Z,Y,X] permuted view of the [X,Y,Z]
        buffer (get_tsdf.py:95-97) and the float64 bounds tensor -- still on the device."""
        X, Y, Z = self._tsdf.shape
        return self._tsdf.reshape(1, 1, X, Y, Z).permute(0, 1, 4, 3, 2), torch.from_numpy(self._vol_bnds.copy())
