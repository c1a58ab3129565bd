"""
Synthetic inputs for tests and bench.py: no datasets, TSDF volumes or
pretrained weights exist in the build container or on the GPU box, so every measured
workload is a seeded "box room" with the bounds / intrinsics of the reference's configs.

Nothing here is on the hot path; it only produces tensors shaped like the ones
DF_Prior.__init__ (reference src/DF_Prior.py:50-116) hands to the Renderer.
"""
import math
import torch

# mapping.bound of the reference's scene configs (configs/Replica/room0.yaml:3,
# configs/Replica/office0.yaml:3, configs/ScanNet/scene0050.yaml:3) + the synthetic stress cube.
SCENE_BOUNDS = {
    'room0': [[-2.9, 8.9], [-3.2, 5.5], [-3.5, 3.3]],
    'office0': [[-5.5, 5.9], [-6.7, 5.4], [-4.7, 5.3]],
    'scene0050': [[0.5, 7.0], [0.0, 4.5], [-0.5, 3.0]],
    'cube16': [[0.0, 16.0 - 0.32], [0.0, 16.0 - 0.32], [0.0, 16.0 - 0.32]],
    'tiny': [[-1.0, 1.8], [-1.2, 1.5], [-0.9, 1.4]],
}


def scene_bound(bound, bound_divisible=0.32, scale=1.0):
    """Round the upper bound up to a multiple of bound_divisible (src/DF_Prior.py:185-190)."""
    b = torch.tensor(bound, dtype=torch.float64) * scale
    b[:, 1] = (((b[:, 1] - b[:, 0]) / bound_divisible).int() + 1) * bound_divisible + b[:, 0]
    return b


def grid_shape(bound, grid_len, c_dim=32):
    """[1, c_dim, Z, Y, X] as src/DF_Prior.py:243-246."""
    xyz_len = bound[:, 1] - bound[:, 0]
    s = list(map(int, (xyz_len / grid_len).tolist()))
    s[0], s[2] = s[2], s[0]
    return [1, c_dim, *s]


def make_grids(bound, low=0.32, high=0.16, color=0.16, c_dim=32, seed=0, device='cpu', std_scale=1.0):
    """Feature grids initialised like src/DF_Prior.py:247-263."""
    g = torch.Generator().manual_seed(seed)
    c = {}
    for key, glen, std in (('grid_low', low, 0.01), ('grid_high', high, 0.0001), ('grid_color', color, 0.01)):
        shp = grid_shape(bound, glen, c_dim)
        c[key] = (torch.randn(shp, generator=g) * std * std_scale).to(device)
    return c


def make_box_room_tsdf(bound, voxel=4.0 / 256, inset=0.6, device='cpu', trunc_voxels=5.0):
    """TSDF of an axis-aligned room whose walls sit `inset` metres inside `bound`.

    Follows the reference's volume conventions: dims = ceil(extent / voxel) and the upper
    bound snapped to dims*voxel (src/fusion.py:42-43), values clamp(sdf / (5*voxel), -1, 1)
    (src/fusion.py:38), physical order [X][Y][Z] exposed as the permuted view
    [1,1,Z,Y,X] (get_tsdf.py:95-97).  Returns (tsdf_view, tsdf_bnds f64 [3,2], inner box).
    """
    b = bound.clone().double()
    dims = torch.ceil((b[:, 1] - b[:, 0]) / voxel).long()
    bnds = b.clone()
    bnds[:, 1] = bnds[:, 0] + dims.double() * voxel
    lo_in = (b[:, 0] + inset).float()
    hi_in = (b[:, 1] - inset).float()
    X, Y, Z = [int(v) for v in dims]
    trunc = trunc_voxels * voxel
    xs = (float(b[0, 0]) + torch.arange(X, device=device, dtype=torch.float32) * voxel)
    ys = (float(b[1, 0]) + torch.arange(Y, device=device, dtype=torch.float32) * voxel)
    zs = (float(b[2, 0]) + torch.arange(Z, device=device, dtype=torch.float32) * voxel)
    dx = torch.minimum(xs - float(lo_in[0]), float(hi_in[0]) - xs)
    dy = torch.minimum(ys - float(lo_in[1]), float(hi_in[1]) - ys)
    dz = torch.minimum(zs - float(lo_in[2]), float(hi_in[2]) - zs)
    vol = torch.empty((X, Y, Z), dtype=torch.float32, device=device)
    # build slab by slab to bound peak memory on 1024^3 volumes
    dyz = torch.minimum(dy[:, None], dz[None, :])
    step = max(1, (1 << 26) // max(1, Y * Z))
    for x0 in range(0, X, step):
        x1 = min(X, x0 + step)
        sdf = torch.minimum(dx[x0:x1, None, None], dyz[None])
        vol[x0:x1] = torch.clamp(sdf / trunc, -1.0, 1.0)
    tsdf = vol.reshape(1, 1, X, Y, Z).permute(0, 1, 4, 3, 2)
    return tsdf, bnds, (lo_in.double(), hi_in.double())


def box_depth(rays_o, rays_d, lo_in, hi_in):
    """Analytic z-depth of the first wall hit from inside the room (rays_d has camera z = -1, so
    the ray parameter IS the sensor depth, cf. src/common.py:254-272)."""
    o = rays_o.double()
    d = rays_d.double()
    t = torch.maximum((lo_in.to(o.device) - o) / d, (hi_in.to(o.device) - o) / d)
    return t.min(dim=-1)[0].float()


def camera_c2w(center, yaw=0.0, pitch=0.0, device='cpu'):
    """Camera-to-world with the reference's convention (camera looks along -z, y up)."""
    cy_, sy_ = math.cos(yaw), math.sin(yaw)
    cp_, sp_ = math.cos(pitch), math.sin(pitch)
    Ry = torch.tensor([[cy_, 0, sy_], [0, 1, 0], [-sy_, 0, cy_]], dtype=torch.float32)
    Rx = torch.tensor([[1, 0, 0], [0, cp_, -sp_], [0, sp_, cp_]], dtype=torch.float32)
    c2w = torch.eye(4, dtype=torch.float32)
    c2w[:3, :3] = Ry @ Rx
    c2w[:3, 3] = torch.tensor(center, dtype=torch.float32)
    return c2w.to(device)


class Scene(object):
    """Bundle of everything a Renderer call needs (mirrors the attributes DF_Prior exposes:
    bound, vol_bnds/tsdf_bnds, shared_c, tsdf_volume_shared, H, W, fx, fy, cx, cy)."""

    def __init__(self, name='room0', H=480, W=640, fx=577.6, fy=577.6, cx=319.5, cy=239.5,
                 voxel=4.0 / 256, device='cpu', seed=0, grid_std_scale=1.0, inset=0.6,
                 low=0.32, high=0.16, color=0.16):
        self.name = name
        self.H, self.W, self.fx, self.fy, self.cx, self.cy = H, W, fx, fy, cx, cy
        self.bound = scene_bound(SCENE_BOUNDS[name])
        self.c = make_grids(self.bound, low, high, color, seed=seed, device=device, std_scale=grid_std_scale)
        self.tsdf_volume, self.tsdf_bnds, (self.lo_in, self.hi_in) = make_box_room_tsdf(
            self.bound, voxel=voxel, inset=inset, device=device)
        self.vol_bnds = self.tsdf_bnds
        self.device = device
        ctr = (self.lo_in + self.hi_in) / 2
        self.center = ctr.tolist()

    def default_c2w(self, offset=(0.0, 0.0, 0.0), yaw=0.3, pitch=-0.1):
        ctr = [self.center[k] + offset[k] for k in range(3)]
        return camera_c2w(ctr, yaw, pitch, self.device)

    def depth_image(self, c2w, zero_band=0.05):
        """gt_depth [H,W] f32: analytic depth + a band of zero-depth (invalid) pixels on the left."""
        from .common import get_rays
        rays_o, rays_d = get_rays(self.H, self.W, self.fx, self.fy, self.cx, self.cy, c2w, self.device)
        d = box_depth(rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), self.lo_in, self.hi_in).reshape(self.H, self.W)
        nzero = int(self.W * zero_band)
        if nzero > 0:
            d[:, :nzero] = 0.0
        return d


SCENE_BOUNDS['mini'] = [[-0.8, 0.7], [-0.6, 0.8], [-0.5, 0.6]]


def mini_scene(device='cpu', seed=0, grid_std_scale=30.0):
    """A ~1.6 m room small enough for committed fixtures: grids 5x5x4 / 10x10x8 voxels, TSDF
    40x40x32.  Feature grids are scaled up so that occupancies are not all ~0."""
    sc = Scene.__new__(Scene)
    sc.name = 'mini'
    sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy = 48, 64, 57.76, 57.76, 31.5, 23.5
    sc.bound = scene_bound(SCENE_BOUNDS['mini'])
    sc.c = make_grids(sc.bound, seed=seed, device=device, std_scale=grid_std_scale)
    sc.c['grid_high'] = sc.c['grid_high'] * 100.0      # N(0,1e-4) would make `high` a no-op
    sc.tsdf_volume, sc.tsdf_bnds, (sc.lo_in, sc.hi_in) = make_box_room_tsdf(
        sc.bound, voxel=0.04, inset=0.25, device=device, trunc_voxels=3.0)
    sc.vol_bnds = sc.tsdf_bnds
    sc.device = device
    sc.center = ((sc.lo_in + sc.hi_in) / 2).tolist()
    return sc


def make_ray_batch(scene, n_rays, seed=1, zero_frac=0.1, noise=0.02, poses=2, device='cpu'):
    """Seeded ray batch in the shape src/Mapper.py:407-435 assembles: random pixels of a few
    poses inside the room, analytic depth (+noise), a fraction of zero-depth pixels, random colour."""
    from .common import get_rays
    g = torch.Generator().manual_seed(seed)
    ro_l, rd_l, d_l = [], [], []
    per = [n_rays // poses + (1 if k < n_rays % poses else 0) for k in range(poses)]
    for k in range(poses):
        off = ((torch.rand(3, generator=g) - 0.5) * 0.3).tolist()
        yaw = float(torch.rand(1, generator=g)) * 6.28
        pitch = (float(torch.rand(1, generator=g)) - 0.5) * 0.8
        c2w = scene.default_c2w(offset=off, yaw=yaw, pitch=pitch).cpu()
        ro, rd = get_rays(scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy, c2w, 'cpu')
        pick = torch.randint(scene.H * scene.W, (per[k],), generator=g)
        ro = ro.reshape(-1, 3)[pick].float().contiguous()
        rd = rd.reshape(-1, 3)[pick].float().contiguous()
        d = box_depth(ro, rd, scene.lo_in, scene.hi_in)
        d = d * (1.0 + noise * (torch.rand(d.shape, generator=g) * 2 - 1))
        ro_l.append(ro)
        rd_l.append(rd)
        d_l.append(d.float())
    rays_o, rays_d, depth = torch.cat(ro_l), torch.cat(rd_l), torch.cat(d_l)
    zero = torch.rand(n_rays, generator=g) < zero_frac
    depth = torch.where(zero, torch.zeros_like(depth), depth)
    color = torch.rand(n_rays, 3, generator=g)
    return rays_o.to(device), rays_d.to(device), depth.to(device), color.to(device)


def seeded_state_dict(seed=0, bias_scale=0.05, occ_bias=-0.5, out_scale=0.15):
    """Seeded decoder weights (pretrained/low_high.pt is missing from the reference snapshot):
    the reference's init distributions (xavier-uniform DenseLayers, N(0,25^2) Fourier matrices,
    nn.Linear-default fc_c) with small non-zero biases; the occupancy heads are damped and biased
    negative so that free space is mostly transparent, as with trained decoders."""
    from .decoder import DF
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in DF().state_dict().items():
        shp = tuple(v.shape)
        if k.endswith('_B'):
            sd[k] = torch.randn(shp, generator=g) * 25
        elif k.endswith('weight'):
            fan_out, fan_in = shp
            gain = 1.0 if 'output_linear' in k else 2 ** 0.5
            a = (1.0 / fan_in) ** 0.5 if 'fc_c' in k else gain * (6.0 / (fan_in + fan_out)) ** 0.5
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * a
        else:
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * bias_scale
    for name in ('low', 'high'):
        sd[f'{name}_decoder.output_linear.weight'] *= out_scale
    sd['low_decoder.output_linear.bias'] += occ_bias
    sd['color_decoder.output_linear.weight'][3] *= out_scale
    return sd
