"""
bench.py -- rendered rays/s of the per-ray volume-rendering hot path on MI355X.

Workload (BASELINE.json configs[1]): Replica room0-sized synthetic box room, 640x480 full-frame
``Renderer.render_img`` (stage color, no_grad), 64 samples/ray = N_samples 48 + N_surface 16,
in the reference's 100 000-ray batches.  One "step" = one full frame (307 200 rays).  Inputs
(grids, 785 MB TSDF, decoder weights, pose, depth image) are resident in HBM before the timed
region; the per-call layout conversions of my path (grid relayout, weight packing) are
invalidated every step so that they are INSIDE the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N   (one rank per GPU)

Multi-GPU: rays (here: whole frames, one pose per rank) are independent units, so ranks render
with NO data-path collective ("weak" scaling); value = all rays of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      dominant kernel (colour decoder, MFMA f32 bound): algorithmic FLOP per launch /
                HIP-event time of that kernel alone on the launch stream
  roofline_tsdf the HBM-streaming trilerp stage (k_tsdf): 32 algorithmic bytes per sample
  cpu_baseline  the oracle (CPU PyTorch restatement == reference) timed on this host's cores on a
                bounded ray sample of the same workload; psnr/max-rel of the GPU path vs it
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic work per sample point (SURVEY.md section 8d / BASELINE.md section 3)
MAC_LOW, MAC_HIGH, MAC_COLOR, MAC_ATT = 15479, 20599, 15575, 33024
TSDF_BYTES_PER_SAMPLE = 32
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, spec
PEAK_F16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense f16/bf16 MFMA, spec
F16X3_FLOP_COLOR = 90 * 32 * 32 * 16 * 2 / 32.0   # executed f16 MFMA FLOP per sample: 90 x 32x32x16 per 32-point tile
PEAK_HBM_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec


def pmc_traffic(kernel_prefix, samples_per_launch):
    """HBM bytes per launch of one kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE, separate runs, profiles/r01_pmc_hbm_traffic.csv holds bytes per sample): the counters
    cannot be read from inside this process, so the figure is the profiled bytes/sample x this run's samples
    per launch.  None when the file is absent."""
    import csv
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_hbm_traffic.csv')
    if not os.path.exists(path):
        return None
    with open(path) as f:
        rows = [r for r in csv.reader(l for l in f if not l.startswith('#'))]
    for r in rows[1:]:
        if r[0].startswith(kernel_prefix):
            return (float(r[3]) + float(r[4])) * samples_per_launch
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--scene', default='room0')
    ap.add_argument('--cpu-rays', type=int, default=20000, help='ray sample of the CPU baseline leg (0 = skip)')
    ap.add_argument('--no-stage-timing', action='store_true')
    return ap.parse_args()


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 or os.environ.get('ADFP_BENCH_FORCE_DIST') == '1':     # the env knob exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device(f'cuda:{local_rank}'))
    else:
        dist = None
    n_gpus = world
    if args.gpus != world and rank == 0:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run '
              f'--nproc-per-node {args.gpus} for a {args.gpus}-GPU run (reporting n_gpus={world})', file=sys.stderr)
    dev = torch.device(f'cuda:{local_rank}')
    torch.cuda.set_device(dev)

    import attentive_dfprior_amd as A
    from attentive_dfprior_amd import synthetic, _lib
    L = _lib.lib()

    H, W, NS, NF = 480, 640, 48, 16
    S = NS + NF
    scene = synthetic.Scene(args.scene, H=H, W=W, device=dev, grid_std_scale=20.0)
    scene.c['grid_high'] = scene.c['grid_high'] * 100
    sd = synthetic.seeded_state_dict(0)
    dec = A.DF()
    dec.load_state_dict(sd)
    dec.bound = scene.bound
    dec = dec.to(dev)
    cfg = {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': NS, 'N_surface': NF, 'N_importance': 0},
           'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}
    rend = A.Renderer(cfg, None, scene)
    tsdf_bnds = scene.tsdf_bnds.to(dev)
    # one pose per rank (weak scaling: every rank renders a full frame)
    c2w = scene.default_c2w(offset=(0.3 * rank, 0.1 * rank, 0.0), yaw=0.3 + 0.4 * rank, pitch=-0.1)
    gt_depth = scene.depth_image(c2w)
    n_rays = H * W

    def step():
        rend._engine._grid_cache.clear()      # relayout + packing inside the timed region
        dec._packed.clear()
        return rend.render_img(scene.c, dec, c2w, dev, scene.tsdf_volume, tsdf_bnds, 'color', gt_depth=gt_depth)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    depth_img, unc_img, color_img = out
    assert torch.isfinite(depth_img).all() and torch.isfinite(color_img).all()

    ms_per_step = elapsed / args.steps * 1e3
    value = n_gpus * n_rays * args.steps / elapsed

    result = {
        'metric': 'rendered rays/sec (64 samples/ray), Replica room0',
        'value': value, 'unit': 'rays/s', 'n_gpus': n_gpus, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic', 'math': os.environ.get('ADFP_MATH', 'f16x3'),
        'config': {'workload': f'{args.scene} synthetic box room, 640x480 full-frame render_img, stage color, '
                               '64 samples/ray (N_samples 48 + N_surface 16), ray_batch_size 100000, '
                               'one frame per GPU per step',
                   'rays_per_step_per_gpu': n_rays, 'samples_per_ray': S,
                   'tsdf_voxels': list(scene.tsdf_volume.shape[2:]),
                   'grid_high': list(scene.c['grid_high'].shape[2:])},
    }

    if rank == 0 and not args.no_stage_timing:
        result.update(stage_timing(L, _lib, rend, dec, scene, tsdf_bnds, c2w, gt_depth, dev, NS, NF))
        if os.environ.get('ADFP_MATH', 'f16x3') == 'f16x3':
            # the same frame with every product on the exact f32-input MFMA (ADFP_MATH=f32), for reference:
            # the f16x3 split reproduces f32 products to 2^-22, this is the bit-exact mode
            os.environ['ADFP_MATH'] = 'f32'
            try:
                step()
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(3):
                    out32 = step()
                torch.cuda.synchronize(dev)
                dt = (time.perf_counter() - t1) / 3
            finally:
                os.environ['ADFP_MATH'] = 'f16x3'
            result['exact_f32_mode'] = {
                'value': n_rays / dt, 'unit': 'rays/s', 'ms_per_step': dt * 1e3, 'n_gpus': 1,
                'max_abs_diff_color_vs_default_mode': float((out32[2] - color_img).abs().max()),
                'max_rel_diff_depth_vs_default_mode': float(((out32[0] - depth_img).abs() / depth_img.abs().clamp_min(1e-3)).max())}
    if rank == 0 and args.cpu_rays > 0:
        result.update(cpu_leg(rend, dec, sd, scene, tsdf_bnds, c2w, gt_depth, dev, NS, NF, args.cpu_rays))
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def stage_timing(L, _lib, rend, dec, scene, tsdf_bnds, c2w, gt_depth, dev, NS, NF, reps=5):
    """HIP-event time of the individual kernels, each launched alone on torch's current stream
    (the stream the library launches on), over EXACTLY the launch mix of one frame: the
    reference's ray batches (3 x 100 000 + 1 x 7 200 rays at 640x480), so that the average launch
    duration equals what `rocprofv3 --kernel-trace --stats` reports for the same command."""
    from attentive_dfprior_amd.common import get_rays
    eng = rend._engine
    S = NS + NF
    ro_all, rd_all = get_rays(scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy, c2w, dev)
    ro_all, rd_all, gd_all = ro_all.reshape(-1, 3), rd_all.reshape(-1, 3), gt_depth.reshape(-1)
    sc, keep = eng.scene(dec, scene.c, scene.tsdf_volume, tsdf_bnds, scene.bound, 'color')
    batches, n_band, n_pts = [], 0, 0
    for i in range(0, ro_all.shape[0], rend.ray_batch_size):
        ro = ro_all[i:i + rend.ray_batch_size].contiguous()
        rd = rd_all[i:i + rend.ray_batch_size].contiguous()
        gd = gd_all[i:i + rend.ray_batch_size].contiguous()
        with torch.no_grad():
            d, u, c, w, aux = eng.render_forward(dec, scene.c, ro, rd, gd, scene.tsdf_volume, tsdf_bnds, scene.bound,
                                                 'color', NS, NF, want_aux=True)
        P = ro.shape[0] * S
        n_band += int((w != 1).sum())
        n_pts += P
        ap = _lib.AdfpPoints()
        ap.mode = _lib.PTS_RAYS
        ap.n_points = P
        ap.rays_o, ap.rays_d, ap.z_vals, ap.S = ro.data_ptr(), rd.data_ptr(), aux['z_vals'].data_ptr(), S
        batches.append((ap, ro, rd, gd, aux['z_vals']))
    Pmax = max(b[0].n_points for b in batches)
    raw = torch.empty((Pmax, 4), dtype=torch.float32, device=dev)
    wbuf = torch.empty((Pmax,), dtype=torch.float32, device=dev)
    flags = torch.empty((Pmax,), dtype=torch.uint8, device=dev)
    lst = torch.empty((Pmax,), dtype=torch.int32, device=dev)
    attu = torch.empty((Pmax,), dtype=torch.float32, device=dev)
    cnt = torch.zeros((4,), dtype=torch.int32, device=dev)
    st = _lib.current_stream(dev)
    band_frac = n_band / n_pts
    nl = len(batches)

    def timed(fn):
        """average duration of ONE launch (s) over reps x the frame's launch mix"""
        for b in batches:
            fn(b)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            for b in batches:
                fn(b)
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / (reps * nl) * 1e-3

    t_color = timed(lambda b: _lib.check(L.adfp_decode_stage(C.byref(sc), C.byref(b[0]), 2, _lib.ptr(raw), _lib.ptr(wbuf), st), 'decode'))
    t_low = timed(lambda b: _lib.check(L.adfp_decode_stage(C.byref(sc), C.byref(b[0]), 0, _lib.ptr(raw), _lib.ptr(wbuf), st), 'decode'))
    t_tsdf = timed(lambda b: _lib.check(L.adfp_tsdf_stage(C.byref(sc), C.byref(b[0]), _lib.ptr(flags), _lib.ptr(lst), _lib.ptr(attu),
                                                          None, _lib.ptr(cnt), st), 'tsdf'))
    t_all = timed(lambda b: eng.render_forward(dec, scene.c, b[1], b[2], b[3], scene.tsdf_volume, tsdf_bnds, scene.bound,
                                               'color', NS, NF))
    pts_per_launch = n_pts / nl
    fl_color = 2.0 * MAC_COLOR * pts_per_launch
    ach = fl_color / t_color / 1e12
    by = float(TSDF_BYTES_PER_SAMPLE) * pts_per_launch
    useful = 2.0 * (MAC_LOW + MAC_COLOR + band_frac * (MAC_HIGH + MAC_ATT)) * pts_per_launch
    from attentive_dfprior_amd.engine import math_mode
    if math_mode() == 'f32':
        roof = {'kernel': 'k_decode<32,4,COLOR> (colour decoder, exact f32-input MFMA)', 'bound': 'mfma',
                'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_F32_MFMA_TFLOPS,
                'traffic': None}
    else:
        ex = F16X3_FLOP_COLOR * pts_per_launch / t_color / 1e12
        roof = {'kernel': 'k_decode_h<32,4,COLOR> (colour decoder, f16 MFMA with 3-product f32 operand split)',
                'bound': 'mfma', 'achieved': ex, 'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': ex / PEAK_F16_MFMA_TFLOPS, 'traffic': None,
                'note': 'executed f16 MFMA FLOP (3 products per f32 product) against the 2.4 GHz spec peak; the chip '
                        'clocks this kernel at ~1.8 GHz and PMC shows the matrix pipe busy 41 % of the time '
                        '(profiles/r01_pmc_sq_forward.csv); the rest of the issue slots go to the gather / Fourier / split '
                        'VALU work, which barely overlaps MFMA on a CDNA4 SIMD (tools/micro, DESIGN.md sections 4.1, 5)',
                'algorithmic_f32_tflops': ach, 'f32_mfma_peak_tflops': PEAK_F32_MFMA_TFLOPS,
                'frac_of_f32_mfma_peak': ach / PEAK_F32_MFMA_TFLOPS}
    kname = 'void k_decode<32, 4, 2' if math_mode() == 'f32' else 'void k_decode_h<32, 4, 2'
    roof['traffic'] = pmc_traffic(kname, pts_per_launch)
    roof['traffic_note'] = ('HBM bytes per launch = (FETCH_SIZE + WRITE_SIZE) per sample of profiles/r01_pmc_hbm_traffic.csv '
                            '(separate --pmc passes) x samples per launch; stores are 16-B/lane streams (exact), '
                            'gather fetches are uncalibrated on gfx950 (MI355X_MICROARCH.md, HBM)')
    roof.update({'algorithmic_flop_per_launch': fl_color, 'avg_launch_ms': t_color * 1e3, 'launches_per_frame': nl,
                 'points_per_launch': pts_per_launch})
    return {
        'roofline': roof,
        'roofline_tsdf': {'kernel': 'k_tsdf (TSDF trilerp + band mask + compaction)', 'bound': 'hbm',
                          'achieved': by / t_tsdf / 1e9, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s',
                          'frac': by / t_tsdf / 1e9 / PEAK_HBM_GBPS, 'traffic': pmc_traffic('k_tsdf', pts_per_launch),
                          'bytes_per_launch': by,
                          'avg_launch_ms': t_tsdf * 1e3},
        'stage_avg_launch_ms': {'color_decoder': t_color * 1e3, 'low_decoder': t_low * 1e3, 'tsdf': t_tsdf * 1e3,
                                'whole_render_batch_ray': t_all * 1e3},
        'in_band_fraction': band_frac,
        'useful_tflops_whole_frame': useful / t_all / 1e12,
    }


def cpu_leg(rend, dec, sd, scene, tsdf_bnds, c2w, gt_depth, dev, NS, NF, n_cpu):
    """The oracle on this host's cores on a bounded ray sample of the SAME workload, and the
    GPU result on exactly those rays for PSNR / max-rel."""
    from oracle import adfp_oracle as O
    from attentive_dfprior_amd.common import get_rays
    ro, rd = get_rays(scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy, c2w, dev)
    tot = scene.H * scene.W
    pick = torch.arange(0, tot, max(1, tot // n_cpu), device=dev)[:n_cpu]
    ro, rd, gd = ro.reshape(-1, 3)[pick].contiguous(), rd.reshape(-1, 3)[pick].contiguous(), gt_depth.reshape(-1)[pick].contiguous()
    with torch.no_grad():
        d, u, c, w = rend.render_batch_ray(scene.c, dec, rd, ro, dev, scene.tsdf_volume, tsdf_bnds, 'color', gt_depth=gd)
    ncpu = os.cpu_count() or 1
    c_cpu = {k: v.cpu() for k, v in scene.c.items()}
    tsdf_cpu = scene.tsdf_volume.cpu()
    ro_c, rd_c, gd_c = ro.cpu(), rd.cpu(), gd.cpu()
    # thread count: big hosts oversubscribe badly with all cores, so pick the faster of {all cores, 32}
    # on a 2 000-ray probe, then time the whole sample with it
    probe = slice(0, min(2000, ro_c.shape[0]))
    cores, tprobe = ncpu, None
    for threads in sorted({ncpu, min(32, ncpu)}):
        torch.set_num_threads(threads)
        with torch.no_grad():
            t0 = time.perf_counter()
            O.render_batch_ray(sd, c_cpu, rd_c[probe], ro_c[probe], tsdf_cpu, scene.tsdf_bnds, scene.bound, 'color',
                               gd_c[probe], NS, NF)
            dt = time.perf_counter() - t0
        if tprobe is None or dt < tprobe:
            tprobe, cores = dt, threads
    torch.set_num_threads(cores)
    best = None
    with torch.no_grad():
        for it in range(3):          # 1 warm-up + best of 2
            t0 = time.perf_counter()
            od, ou, oc, ow = O.render_batch_ray(sd, c_cpu, rd_c, ro_c, tsdf_cpu, scene.tsdf_bnds, scene.bound,
                                                'color', gd_c, NS, NF)
            dt = time.perf_counter() - t0
            if it > 0 and (best is None or dt < best):
                best = dt
    # forward + backward of the Mapper loss (src/Mapper.py:457-473) on a 2 000-ray subset: the oracle with autograd on
    # the host cores beside the product path's autograd on the GPU, same rays, gradients to grids + colour / attention nets
    nt = min(2000, ro_c.shape[0])
    gcol = torch.rand(nt, 3, generator=torch.Generator().manual_seed(0))
    train = {}
    try:
        c_req = {k: v.clone().requires_grad_(True) for k, v in c_cpu.items()}
        sd_req = {k: (v.clone().requires_grad_(True) if k.startswith(('color_decoder', 'mlp')) else v) for k, v in sd.items()}
        tb = None
        for it in range(2):
            for v in list(c_req.values()) + [v for v in sd_req.values() if v.requires_grad]:
                v.grad = None
            t0 = time.perf_counter()
            td, tu, tc, tw = O.render_batch_ray(sd_req, c_req, rd_c[:nt], ro_c[:nt], tsdf_cpu, scene.tsdf_bnds, scene.bound,
                                                'color', gd_c[:nt], NS, NF)
            O.mapper_loss(td, tc, tw, gd_c[:nt], gcol, 'color').backward()
            dt = time.perf_counter() - t0
            tb = dt if tb is None or dt < tb else tb
        c_g = {k: v.detach().clone().requires_grad_(True) for k, v in scene.c.items()}
        ro_t, rd_t, gd_t, gc_t = ro[:nt], rd[:nt], gd[:nt], gcol.to(dev)
        for p_ in list(dec.low_decoder.parameters()) + list(dec.high_decoder.parameters()):
            p_.requires_grad_(False)

        def gpu_it():
            for v in c_g.values():
                v.grad = None
            dd, uu, cc, ww = rend.render_batch_ray(c_g, dec, rd_t, ro_t, dev, scene.tsdf_volume, tsdf_bnds, 'color', gt_depth=gd_t)
            m = gd_t > 0
            (torch.abs(gd_t[m] - dd[m]).sum() + 0.2 * torch.abs(gc_t - cc).sum()).backward()
        for _ in range(3):
            gpu_it()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(10):
            gpu_it()
        torch.cuda.synchronize(dev)
        tg = (time.perf_counter() - t0) / 10
        gl = c_g['grid_color'].grad.cpu()
        train = {'train_fwd_bwd': {'rays': nt, 'unit': 'rays/s', 'value': nt / tg, 'ms_per_iteration': tg * 1e3,
                                   'cpu_baseline': nt / tb, 'cpu_cores': cores,
                                   'max_rel_grad_grid_color_vs_oracle': float((gl - c_req['grid_color'].grad).abs().max()
                                                                              / c_req['grid_color'].grad.abs().max())}}
        for p_ in dec.parameters():
            p_.grad = None
    except Exception as e:                      # the training leg is extra information, never the reason a bench fails
        train = {'train_fwd_bwd': {'error': repr(e)[:200]}}
    mse = float(((c.cpu().double() - oc.double()) ** 2).mean())
    peak = float(oc.abs().max())
    psnr = 10.0 * torch.log10(torch.tensor(peak * peak / max(mse, 1e-300))).item()
    dmse = float(((d.cpu() - od) ** 2).mean())
    dpeak = float(od.abs().max())
    return {
        'cpu_baseline': {'value': len(pick) / best, 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
                         'sample': f'{len(pick)} evenly strided rays of the same frame, 64 samples/ray, stage color, '
                                   f'no_grad, torch CPU with {cores} threads (faster of {ncpu} / {min(32, ncpu)} on a probe), best of 2 after 1 warm-up '
                                   f'({best:.2f} s)'},
        'parity_vs_oracle': {'psnr_color_db': psnr, 'psnr_depth_db': 10.0 * torch.log10(torch.tensor(dpeak * dpeak / max(dmse, 1e-300))).item(),
                             'max_rel_depth': float(((d.cpu() - od).abs().max() / od.abs().max())),
                             'max_rel_color': float(((c.cpu() - oc).abs().max() / oc.abs().max())),
                             'rays': len(pick)},
        **train,
    }


if __name__ == '__main__':
    main()
