"""Where the wall time of the UNCHANGED callers goes (src/Mapper.py:380-482 / src/Tracker.py:75-134 as they stand: render_batch_ray
under autograd + torch loss + loss.backward() + torch.optim.Adam).  Sections are timed on the host clock; every section is timed
twice: as the caller runs it (asynchronous launches: the HOST cost) and with a device synchronisation after it (host + GPU).

  python tools/host_breakdown.py [--rays 1000 5000] [--iters 200] [--cprofile]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import attentive_dfprior_amd as A                                    # noqa: E402
from attentive_dfprior_amd import synthetic                          # noqa: E402
from attentive_dfprior_amd.common import get_rays                    # noqa: E402


class _StubRender(torch.autograd.Function):
    """What is left of the iteration when render_batch_ray costs NOTHING: outputs and gradients are uninitialised allocations of
    the right shapes (no kernel of ours, no C call), wired into autograd exactly like the real function (rays, three grids, every
    trainable parameter).  The iteration timed with this stub is the floor the CALLER's own torch code sets: zero_grad, the loss
    ops with their boolean-index syncs, the autograd engine with one AccumulateGrad per parameter, torch.optim.Adam."""

    @staticmethod
    def forward(ctx, n, s, rays_o, rays_d, g0, g1, g2, *params):
        dev = g0.device
        ctx.shapes = [g0.shape, g1.shape, g2.shape]
        ctx.params = params
        ctx.set_materialize_grads(False)
        return (torch.empty((n,), dtype=torch.float64, device=dev), torch.empty((n,), dtype=torch.float64, device=dev),
                torch.empty((n, 3), device=dev), torch.empty((n, s, 1), device=dev))

    @staticmethod
    def backward(ctx, *g):
        dev = ctx.params[0].device if ctx.params else g[0].device
        out = [None, None, None, None] + [torch.empty(tuple(sh), device=dev) for sh in ctx.shapes]
        needs = ctx.needs_input_grad
        if needs[2]:
            out[2] = torch.empty((g[0].shape[0], 3), device=dev)
        if needs[3]:
            out[3] = torch.empty((g[0].shape[0], 3), device=dev)
        for k in range(3):
            if not needs[4 + k]:
                out[4 + k] = None
        flat = torch.empty((sum(p.numel() for p in ctx.params),), device=dev) if ctx.params else None
        off = 0
        for p in ctx.params:
            out.append(flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        return tuple(out)


def stub_render_batch_ray(rend, dec):
    def call(c, decoders, rays_d, rays_o, device, tsdf_volume, tsdf_bnds, stage, gt_depth=None):
        params = [p for p in dec.parameters() if p.requires_grad]
        return _StubRender.apply(rays_o.shape[0], rend.N_samples + rend.N_surface, rays_o, rays_d, c['grid_low'], c['grid_high'],
                                 c['grid_color'], *params)
    return call


class Sections(object):
    def __init__(self, sync):
        self.sync, self.t, self.acc, self.n = sync, None, {}, 0

    def start(self):
        if self.sync:
            torch.cuda.synchronize()
        self.t = time.perf_counter()

    def mark(self, name):
        if self.sync:
            torch.cuda.synchronize()
        now = time.perf_counter()
        self.acc[name] = self.acc.get(name, 0.0) + (now - self.t)
        self.t = now

    def report(self, iters, title):
        tot = sum(self.acc.values())
        print(f'  {title}: {tot / iters * 1e3:.3f} ms per iteration')
        for k, v in self.acc.items():
            print(f'    {k:34s} {v / iters * 1e6:8.1f} us')


def mapper_case(n_rays, ns, nf, iters, cprof, floor=True):
    dev = torch.device('cuda:0')
    scene = synthetic.Scene('room0', device=dev, grid_std_scale=20.0)
    scene.c['grid_high'] = scene.c['grid_high'] * 100
    dec = A.DF()
    dec.load_state_dict(synthetic.seeded_state_dict(0))
    dec.bound = scene.bound
    dec = dec.to(dev)
    for p in list(dec.low_decoder.parameters()) + list(dec.high_decoder.parameters()):
        p.requires_grad_(False)
    cfg = {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': ns, 'N_surface': nf, 'N_importance': 0},
           'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}
    rend = A.Renderer(cfg, None, scene)
    tb = scene.tsdf_bnds.to(dev)
    c2w = scene.default_c2w()
    gt = scene.depth_image(c2w)
    ro, rd = get_rays(scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy, c2w, dev)
    pick = torch.randint(scene.H * scene.W, (n_rays,), generator=torch.Generator().manual_seed(0)).to(dev)
    ro, rd, gd = ro.reshape(-1, 3)[pick], rd.reshape(-1, 3)[pick], gt.reshape(-1)[pick]
    gc = torch.rand(n_rays, 3, device=dev)
    c = {k: v.clone().requires_grad_(True) for k, v in scene.c.items()}
    params = list(dec.color_decoder.parameters()) + list(dec.mlp.parameters())
    opt = torch.optim.Adam([{'params': params, 'lr': 0.005}, {'params': list(c.values()), 'lr': 0.005}])

    def it(sec=None):
        opt.zero_grad()
        if sec:
            sec.mark('optimizer.zero_grad')
        d, u, col, w = rend.render_batch_ray(c, dec, rd, ro, dev, scene.tsdf_volume, tb, 'color', gt_depth=gd)
        if sec:
            sec.mark('render_batch_ray (forward)')
        m = gd > 0
        loss = torch.abs(gd[m] - d[m]).sum() + 0.2 * torch.abs(gc - col).sum()
        if sec:
            sec.mark('loss (torch ops, 2 bool-index syncs)')
        loss.backward()
        if sec:
            sec.mark('loss.backward()')
        opt.step()
        if sec:
            sec.mark('optimizer.step (torch Adam)')

    for _ in range(10):
        it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        it()
    torch.cuda.synchronize()
    print(f'mapper-shaped iteration {n_rays} rays x {ns + nf}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per iteration (free running)')
    for sync in (False, True):
        sec = Sections(sync)
        for _ in range(iters):
            sec.start()
            it(sec)
        torch.cuda.synchronize()
        sec.report(iters, 'host + GPU per section (synchronised)' if sync else 'host cost per section (asynchronous)')
    t0 = time.perf_counter()
    for _ in range(iters):
        it()
    torch.cuda.synchronize()
    print(f'  free running again: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per iteration')
    if floor:
        real = rend.render_batch_ray
        rend.render_batch_ray = stub_render_batch_ray(rend, dec)
        for _ in range(10):
            it()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            it()
        torch.cuda.synchronize()
        print(f'  FLOOR (render_batch_ray replaced by an allocation-only stub; values are garbage): {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per iteration')
        sec = Sections(False)
        for _ in range(iters):
            sec.start()
            it(sec)
        torch.cuda.synchronize()
        sec.report(iters, 'floor, host cost per section')
        rend.render_batch_ray = real
    from attentive_dfprior_amd import _lib
    if _lib.HOST_TIMING is not None:
        n_it = 10 + 4 * iters
        print('  host time inside the C entry points (ADFP_HOST_TIMING), per iteration:')
        for k, (calls, sec) in sorted(_lib.HOST_TIMING.items(), key=lambda kv: -kv[1][1]):
            print(f'    {k:36s} {calls / n_it:5.1f} calls  {sec / n_it * 1e6:7.1f} us')
        _lib.HOST_TIMING.clear()
    if cprof:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(iters):
            it()
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(40)
        pstats.Stats(pr).sort_stats('cumulative').print_stats('attentive_dfprior_amd|optim/|autograd', 40)
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU]) as prof:
            for _ in range(20):
                it()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=60, max_name_column_width=60))


def tracker_case(n, iters, cprof, floor=True):
    import bench
    import bench_extra as BX
    from attentive_dfprior_amd import common
    from oracle import adfp_oracle as O
    dev = torch.device('cuda:0')
    scene, sd, dec = bench.build_scene(A, synthetic, 'room0', dev)
    for p in dec.parameters():
        p.requires_grad_(False)
    rend = A.Renderer(BX._cfg(48, 16), None, scene)
    tb = scene.tsdf_bnds.to(dev)
    bound = scene.bound.to(dev)
    c2w_gt = scene.default_c2w()
    depth = scene.depth_image(c2w_gt)
    color = torch.rand((scene.H, scene.W, 3), generator=torch.Generator().manual_seed(0)).to(dev)
    H, W, edge = scene.H, scene.W, 20
    cam = BX._tensor_from_c2w(c2w_gt).to(dev)
    cam[4:] += 0.01
    cam.requires_grad_(True)
    opt = torch.optim.Adam([cam], lr=1e-3)

    def it(sec=None):
        opt.zero_grad()
        c2w = common.get_camera_from_tensor(cam)
        if sec:
            sec.mark('zero_grad + camera tensor -> c2w')
        ro, rd, gd, gc = common.get_samples(edge, H - edge, edge, W - edge, n, H, W, scene.fx, scene.fy, scene.cx, scene.cy, c2w, depth, color, dev)
        if sec:
            sec.mark('get_samples')
        ro, rd, gd, gc = common.filter_rays_in_bound(ro, rd, gd, gc, bound)
        if sec:
            sec.mark('pre-filter (1 sync)')
        d, u, col, _ = rend.render_batch_ray(scene.c, dec, rd, ro, dev, scene.tsdf_volume, tb, 'color', gt_depth=gd)
        if sec:
            sec.mark('render_batch_ray (forward)')
        loss = O.tracker_loss(d, u.detach(), col, gd, gc)
        if sec:
            sec.mark('tracker loss (torch ops)')
        loss.backward()
        if sec:
            sec.mark('loss.backward()')
        opt.step()
        if sec:
            sec.mark('optimizer.step (torch Adam)')

    for _ in range(10):
        it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        it()
    torch.cuda.synchronize()
    print(f'tracker-shaped iteration {n} rays x 64: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per iteration (free running)')
    for sync in (False, True):
        sec = Sections(sync)
        for _ in range(iters):
            sec.start()
            it(sec)
        torch.cuda.synchronize()
        sec.report(iters, 'host + GPU per section (synchronised)' if sync else 'host cost per section (asynchronous)')
    t0 = time.perf_counter()
    for _ in range(iters):
        it()
    torch.cuda.synchronize()
    print(f'  free running again: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per iteration')
    if floor:
        real = rend.render_batch_ray
        rend.render_batch_ray = stub_render_batch_ray(rend, dec)
        for _ in range(10):
            it()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            it()
        torch.cuda.synchronize()
        print(f'  FLOOR (render_batch_ray replaced by an allocation-only stub; values are garbage): {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per iteration')
        sec = Sections(False)
        for _ in range(iters):
            sec.start()
            it(sec)
        torch.cuda.synchronize()
        sec.report(iters, 'floor, host cost per section')
        rend.render_batch_ray = real
    from attentive_dfprior_amd import _lib
    if _lib.HOST_TIMING is not None:
        n_it = 10 + 4 * iters
        print('  host time inside the C entry points (ADFP_HOST_TIMING), per iteration:')
        for k, (calls, sec) in sorted(_lib.HOST_TIMING.items(), key=lambda kv: -kv[1][1]):
            print(f'    {k:36s} {calls / n_it:5.1f} calls  {sec / n_it * 1e6:7.1f} us')
        _lib.HOST_TIMING.clear()
    if cprof:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(iters):
            it()
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(40)
        pstats.Stats(pr).sort_stats('cumulative').print_stats('attentive_dfprior_amd|optim/|autograd|oracle|bench_extra', 40)
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU]) as prof:
            for _ in range(20):
                it()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=60, max_name_column_width=60))
    for p in dec.parameters():
        p.requires_grad_(True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rays', type=int, nargs='+', default=[1000, 5000])
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--cprofile', action='store_true')
    ap.add_argument('--no-tracker', action='store_true')
    args = ap.parse_args()
    print('torch', torch.__version__, '| host cores', os.cpu_count())
    for n in args.rays:
        ns, nf = (32, 16) if n <= 1000 else (48, 16)
        mapper_case(n, ns, nf, args.iters, args.cprofile)
    if not args.no_tracker:
        tracker_case(200, args.iters, args.cprofile)


if __name__ == '__main__':
    main()
