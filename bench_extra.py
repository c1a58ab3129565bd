"""
Extra legs of bench.py (rank 0, N = 1): BASELINE.json's other configurations and the built "next" callers of SURVEY.md section
8f, each timed on the HIP path with the oracle on the host cores beside it.  None of them is the headline `value`.

  config1            BASELINE.json configs[0]: room0, 1 000 rays by get_samples (seed 1), N_samples 32 + N_surface 16 --
                     forward (no_grad) and forward + backward of the Mapper loss, HIP path and oracle
  config3            BASELINE.json configs[2]: office0, the 200-frame mapping loop (every 5th frame = 40 mapping calls x 60
                     iterations + a 300-iteration first frame, 5 000 rays) through the fused MapperIteration
  tracker_iteration  one camera-tracking iteration (reference src/Tracker.py:75-134) at 200 and 1 000 rays: get_samples from a
                     quaternion + translation camera tensor, bbox pre-filter, render stage color, the Tracker loss, backward to the
                     camera tensor (ray gradients -> adfp_rays_from_uv_backward), Adam on the pose
  mesher_query       the Mesher's point query (reference src/utils/Mesher.py:437-447): 256^3 = 16.8 M float32 lattice points in
                     500 000-point chunks through eval_points_tsdf + eval_points, stages high and color
  replica_native_frame  the frame the reference itself renders on Replica: 1200 x 680, fx = fy = 600 (configs/Replica/replica.yaml:46-53),
                     N_samples 32 + N_surface 16 (configs/df_prior.yaml:94-95), ray_batch_size 100000 -> 816 000 rays in nine batches,
                     39.2 M sample points, Renderer.render_img under no_grad (src/utils/Renderer.py:278-327)
  allreduce_model    SURVEY.md section 5/8e ring model of the training all-reduce at 2 / 4 / 8 GPUs (a PREDICTION to read the first
                     real multi-GPU record against; nothing here is measured)
"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
MAC_LOW, MAC_HIGH, MAC_COLOR, MAC_ATT = 15479, 20599, 15575, 33024
PEAK_F16_MFMA_TFLOPS, PEAK_F32_MFMA_TFLOPS = 2500.0, 157.3
XGMI_LINK_GBPS = 153.0            # MI355X_MICROARCH.md: per-direction xGMI link, 7 links per GPU


def _cfg(ns, nf):
    return {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': ns, 'N_surface': nf, 'N_importance': 0},
            'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}


def _wall(fn, reps, dev, warm=2):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / reps, out


def _threads_for_oracle(probe):
    """big hosts oversubscribe badly with all cores: the faster of {all cores, 32} on a probe call"""
    import torch
    ncpu = os.cpu_count() or 1
    best, cores = None, ncpu
    for threads in sorted({ncpu, min(32, ncpu)}):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        probe()
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, cores = dt, threads
    torch.set_num_threads(cores)
    return cores


# ----------------------------------------------------------------------------------------------------------------------
def config1_leg(A, synthetic, scene, sd, dec, dev, cores_hint=None):
    """BASELINE.json configs[0] on the bench scene: the Mapper's own batch shape (configs/df_prior.yaml:62, :94-95)."""
    import torch
    from attentive_dfprior_amd import common
    from oracle import adfp_oracle as O
    NS, NF, N = 32, 16, 1000
    rend = A.Renderer(_cfg(NS, NF), None, scene)
    tb = scene.tsdf_bnds.to(dev)
    c2w = scene.default_c2w()
    depth = scene.depth_image(c2w)
    color = torch.rand((scene.H, scene.W, 3), generator=torch.Generator().manual_seed(0)).to(dev)
    torch.manual_seed(1)
    ro, rd, gd, gc = common.get_samples(0, scene.H, 0, scene.W, N, scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy, c2w, depth, color, dev)
    ro, rd = ro.detach(), rd.detach()

    def fwd():
        with torch.no_grad():
            return rend.render_batch_ray(scene.c, dec, rd, ro, dev, scene.tsdf_volume, tb, 'color', gt_depth=gd)
    t_f, out = _wall(fwd, 50, dev, warm=5)
    frozen = list(dec.low_decoder.parameters()) + list(dec.high_decoder.parameters())
    for p in frozen:
        p.requires_grad_(False)
    cg = {k: v.detach().clone().requires_grad_(True) for k, v in scene.c.items()}

    def fwd_bwd():
        for v in cg.values():
            v.grad = None
        d, u, col, w = rend.render_batch_ray(cg, dec, rd, ro, dev, scene.tsdf_volume, tb, 'color', gt_depth=gd)
        O.mapper_loss(d, col, w, gd, gc, 'color').backward()
    try:
        t_b, _ = _wall(fwd_bwd, 30, dev, warm=5)
        g_color = cg['grid_color'].grad.detach().cpu()
    finally:
        for p in dec.parameters():
            p.grad = None
        for p in frozen:
            p.requires_grad_(True)
    # the oracle on the host cores, the same 1 000 rays
    c_cpu = {k: v.cpu() for k, v in scene.c.items()}
    tsdf_cpu = scene.tsdf_volume.cpu()
    ro_c, rd_c, gd_c, gc_c = ro.cpu(), rd.cpu(), gd.cpu(), gc.cpu()

    def oracle_fwd():
        with torch.no_grad():
            return O.render_batch_ray(sd, c_cpu, rd_c, ro_c, tsdf_cpu, scene.tsdf_bnds, scene.bound, 'color', gd_c, NS, NF)
    cores = _threads_for_oracle(oracle_fwd)
    t_of = min(_cpu_time(oracle_fwd) for _ in range(3))
    od, ou, oc, ow = oracle_fwd()
    c_req = {k: v.clone().requires_grad_(True) for k, v in c_cpu.items()}
    sd_req = {k: (v.clone().requires_grad_(True) if k.startswith(('color_decoder', 'mlp')) else v) for k, v in sd.items()}

    def oracle_fwd_bwd():
        for v in list(c_req.values()) + [v for v in sd_req.values() if v.requires_grad]:
            v.grad = None
        d, u, col, w = O.render_batch_ray(sd_req, c_req, rd_c, ro_c, tsdf_cpu, scene.tsdf_bnds, scene.bound, 'color', gd_c, NS, NF)
        O.mapper_loss(d, col, w, gd_c, gc_c, 'color').backward()
    t_ob = min(_cpu_time(oracle_fwd_bwd) for _ in range(2))
    d, u, col, w = out
    return {'workload': 'BASELINE.json configs[0]: room0, 1 000 rays by common.get_samples (torch.manual_seed(1)), N_samples 32 + '
                        'N_surface 16 = 48 samples/ray, stage color (configs/df_prior.yaml:62, :94-95)',
            'forward': {'value': N / t_f, 'unit': 'rays/s', 'ms': t_f * 1e3,
                        'cpu_baseline': {'value': N / t_of, 'unit': 'rays/s', 'cores': cores, 'kind': 'port', 'ms': t_of * 1e3}},
            'forward_backward': {'value': N / t_b, 'unit': 'rays/s', 'ms': t_b * 1e3,
                                 'what': 'render_batch_ray under autograd + Mapper loss + backward: grids dense, colour decoder and attention net',
                                 'cpu_baseline': {'value': N / t_ob, 'unit': 'rays/s', 'cores': cores, 'kind': 'port', 'ms': t_ob * 1e3}},
            'parity_vs_oracle': {'max_rel_depth': float((d.cpu() - od).abs().max() / od.abs().max()),
                                 'max_rel_color': float((col.cpu() - oc).abs().max() / oc.abs().max()),
                                 'max_abs_attention_weight': float((w.cpu() - ow).abs().max()),
                                 'max_rel_grad_grid_color': float((g_color - c_req['grid_color'].grad).abs().max()
                                                                  / c_req['grid_color'].grad.abs().max())}}


def _cpu_time(fn):
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


# ----------------------------------------------------------------------------------------------------------------------
def config3_leg(dev):
    """BASELINE.json configs[2] at full length through the fused iteration (tools/mapping_loop.py; tests/test_gpu_config3.py
    asserts the same run)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import mapping_loop as ML
    run = ML.MappingRun('office0', rays=5000, total_frames=200, fused=True, device=str(dev))
    calls = list(range(0, 200, 5))
    held = run.heldout_rays(len(calls), stride=5)
    errs = [run.heldout_error(held)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k, f in enumerate(calls):
        run.map_frame(f, 300 if f == 0 else 60, ML.LR_FIRST_FACTOR if f == 0 else 1.0)
        if (k + 1) % 10 == 0:
            torch.cuda.synchronize(dev)
            t_pause = time.perf_counter()
            errs.append(run.heldout_error(held))
            torch.cuda.synchronize(dev)
            t0 += time.perf_counter() - t_pause                      # the held-out renders are not part of the loop
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    finite = all(bool(torch.isfinite(v).all()) for v in run.c.values()) and all(bool(torch.isfinite(p).all()) for p in run.dec.parameters())
    return {'workload': 'BASELINE.json configs[2]: office0-sized scene (1.51 GB TSDF), 200-frame sequence mapped every 5th frame '
                        '(configs/df_prior.yaml:44) = 40 mapping calls x 60 iterations (+ 300 on the first frame), 5 000 rays x 64 '
                        'samples, frustum-masked grids, fused MapperIteration (graph replay); includes per-call frustum masks and '
                        'get_samples', 'iterations': run.n_iter, 'seconds': dt, 'ms_per_iteration': dt / run.n_iter * 1e3,
            'rays_per_s': 5000 * run.n_iter / dt, 'heldout_depth_l1_by_quarter': errs, 'heldout_depth_l1_final': errs[-1],
            'all_finite': finite, 'latched_to_exact': sorted(run.dec._exact_latch)}


# ----------------------------------------------------------------------------------------------------------------------
def _camera_from_tensor(t):
    """quaternion (w, x, y, z) + translation -> [3,4] camera-to-world, differentiable torch ops (the reference keeps this on the
    host side of the renderer too: src/common.py:139-178)."""
    import torch
    q, T = t[:4], t[4:]
    two_s = 2.0 / (q * q).sum()
    qr, qi, qj, qk = q[0], q[1], q[2], q[3]
    R = torch.stack([torch.stack([1 - two_s * (qj ** 2 + qk ** 2), two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr)]),
                     torch.stack([two_s * (qi * qj + qk * qr), 1 - two_s * (qi ** 2 + qk ** 2), two_s * (qj * qk - qi * qr)]),
                     torch.stack([two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), 1 - two_s * (qi ** 2 + qj ** 2)])])
    return torch.cat([R, T[:, None]], 1)


def _tensor_from_c2w(c2w):
    """rotation matrix -> quaternion (w, x, y, z) + translation; the camera looks along a generic direction, so w is not near 0"""
    import torch
    R = c2w[:3, :3].double().cpu()
    w = math.sqrt(max(1e-12, 1.0 + float(R[0, 0] + R[1, 1] + R[2, 2]))) / 2
    q = torch.tensor([w, float(R[2, 1] - R[1, 2]) / (4 * w), float(R[0, 2] - R[2, 0]) / (4 * w), float(R[1, 0] - R[0, 1]) / (4 * w)])
    return torch.cat([q, c2w[:3, 3].double().cpu()]).float()


def tracker_leg(A, synthetic, scene, sd, dec, dev):
    """One camera-tracking iteration (reference src/Tracker.py:75-134), the renderer's second training caller."""
    import torch
    from attentive_dfprior_amd import common
    from oracle import adfp_oracle as O
    NS, NF = 48, 16
    S = NS + NF
    rend = A.Renderer(_cfg(NS, NF), None, scene)
    tb = scene.tsdf_bnds.to(dev)
    bound = scene.bound.to(dev)
    c2w_gt = scene.default_c2w()
    depth = scene.depth_image(c2w_gt)
    color = torch.rand((scene.H, scene.W, 3), generator=torch.Generator().manual_seed(0)).to(dev)
    H, W = scene.H, scene.W
    edge = 20                                                          # tracking.ignore_edge_W / _H, configs/df_prior.yaml:20-21
    for p in dec.parameters():
        p.requires_grad_(False)                                        # the Tracker optimises the pose only (src/Tracker.py:207-211)
    out = {'workload': 'one Tracker.optimize_cam_in_batch iteration (src/Tracker.py:75-134): camera tensor (quaternion + translation) '
                       '-> c2w -> get_samples inside the ignore-edge window -> bbox pre-filter -> render_batch_ray stage color, 64 '
                       'samples/ray -> Tracker loss (handle_dynamic, w_color_loss 0.5) -> backward to the camera tensor -> Adam '
                       '(lr 1e-3); decoders and grids frozen', 'by_batch': {}}
    try:
        for n in (200, 1000):                                          # tracking.pixels: 200 (configs/df_prior.yaml:26), 1000 (ScanNet)
            cam = _tensor_from_c2w(c2w_gt).to(dev)
            cam[4:] += 0.01                                            # start 1 cm off, like a constant-speed initial guess
            cam.requires_grad_(True)
            opt = torch.optim.Adam([cam], lr=1e-3)
            frac = [0.0]

            def it():
                opt.zero_grad()
                c2w = common.get_camera_from_tensor(cam)
                ro, rd, gd, gc = common.get_samples(edge, H - edge, edge, W - edge, n, H, W, scene.fx, scene.fy, scene.cx, scene.cy, c2w, depth, color, dev)
                ro, rd, gd, gc = common.filter_rays_in_bound(ro, rd, gd, gc, bound)
                d, u, col, _ = rend.render_batch_ray(scene.c, dec, rd, ro, dev, scene.tsdf_volume, tb, 'color', gt_depth=gd)
                loss = O.tracker_loss(d, u.detach(), col, gd, gc)
                loss.backward()
                opt.step()
                frac[0] = ro.shape[0]
                return loss
            t_it, loss = _wall(it, 30, dev, warm=5)
            kept = int(frac[0])
            # the same iteration as ONE captured kernel sequence (tracking.TrackerIteration): no autograd engine, no host read-back
            from attentive_dfprior_amd.tracking import TrackerIteration
            fused = TrackerIteration(rend, dec, scene.c, scene.tsdf_volume, tb, H, W, scene.fx, scene.fy, scene.cx, scene.cy, edge, edge)
            cam0 = _tensor_from_c2w(c2w_gt).to(dev)
            cam0[4:] += 0.01
            fused.new_frame(cam0, depth, color)
            t_f, loss_f = _wall(lambda: fused.step(n), 100, dev, warm=5)
            # algorithmic FLOP of the iteration: forward + the input-gradient chain (1x forward) of the four networks on n x S samples
            band = 0.17
            flop = 2.0 * 2.0 * (MAC_LOW + MAC_COLOR + band * (MAC_HIGH + MAC_ATT)) * n * S         # no weight gradients: frozen nets
            # the oracle on the host cores: the same iteration with torch autograd
            cam_c = cam.detach().cpu().clone().requires_grad_(True)
            c_cpu = {k: v.cpu() for k, v in scene.c.items()}
            tsdf_cpu, depth_c, color_c = scene.tsdf_volume.cpu(), depth.cpu(), color.cpu()
            g = torch.Generator().manual_seed(3)

            def oracle_it():
                cam_c.grad = None
                c2w = _camera_from_tensor(cam_c)
                pick = torch.randint((H - 2 * edge) * (W - 2 * edge), (n,), generator=g)
                jj, ii = (pick // (W - 2 * edge) + edge).float(), (pick % (W - 2 * edge) + edge).float()
                ro, rd = O.get_rays_from_uv(ii, jj, c2w, scene.fx, scene.fy, scene.cx, scene.cy)
                gd = depth_c[jj.long(), ii.long()]
                gc = color_c[jj.long(), ii.long()]
                keep = O.prefilter_mask(ro.detach(), rd.detach(), gd, scene.bound)
                ro, rd, gd, gc = ro[keep], rd[keep], gd[keep], gc[keep]
                d, u, col, _ = O.render_batch_ray(sd, c_cpu, rd, ro, tsdf_cpu, scene.tsdf_bnds, scene.bound, 'color', gd, NS, NF)
                O.tracker_loss(d, u.detach(), col, gd, gc).backward()
            cores = _threads_for_oracle(oracle_it)
            t_o = min(_cpu_time(oracle_it) for _ in range(2))
            out['by_batch'][str(n)] = {'rays_sampled': n, 'rays_after_prefilter': kept,
                                       'ms_per_iteration': t_f * 1e3, 'rays_per_s': n / t_f, 'final_loss': float(loss_f),
                                       'algorithmic_tflops': flop / t_f / 1e12,
                                       'frac_of_f32_mfma_peak': flop / t_f / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                       'path': 'tracking.TrackerIteration: one HIP-graph replay per iteration (the n sampled rays are rendered, '
                                               'the pre-filter is a keep flag; the pose gradient takes the exact f32-input MFMA backward)',
                                       'reference_shaped_calls': {'ms_per_iteration': t_it * 1e3, 'final_loss': float(loss.detach()),
                                                                  'note': 'get_samples -> filter_rays_in_bound -> render_batch_ray -> torch loss -> '
                                                                          'backward() -> torch.optim.Adam against this package: bound by host dispatch '
                                                                          'and two read-backs, not by the GPU'},
                                       'cpu_baseline': {'ms_per_iteration': t_o * 1e3, 'rays_per_s': kept / t_o, 'cores': cores, 'kind': 'port'}}
    finally:
        for p in dec.parameters():
            p.requires_grad_(True)
            p.grad = None
    return out


# ----------------------------------------------------------------------------------------------------------------------
def mesher_leg(A, synthetic, scene, sd, dec, dev, resolution=256, chunk=500000):
    """The Mesher's point query (reference src/utils/Mesher.py:365-393 lattice, :437-447 chunk loop, :286-326 eval_points)."""
    import numpy as np
    import torch
    from oracle import adfp_oracle as O
    rend = A.Renderer(_cfg(48, 16), None, scene)
    tb = scene.tsdf_bnds.to(dev)
    b = scene.bound.double()
    pad = 0.05
    axes = [torch.linspace(float(b[k, 0]) - pad, float(b[k, 1]) + pad, resolution, dtype=torch.float64) for k in range(3)]
    xx, yy, zz = torch.meshgrid(axes[0], axes[1], axes[2], indexing='xy')                     # np.meshgrid default
    pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], 1).float().to(dev)   # dtype=torch.float, Mesher.py:387-389
    P = pts.shape[0]
    out = {'workload': f'Mesher.get_mesh point query: {resolution}^3 = {P} float32 lattice points over bound + 0.05 m, in {chunk}-point '
                       'chunks, eval_points_tsdf + eval_points per chunk (src/utils/Mesher.py:437-447)', 'points': P, 'by_stage': {}}
    band = None
    for stage in ('high', 'color'):
        def query():
            res = []
            with torch.no_grad():
                for pi in torch.split(pts, chunk, dim=0):
                    rend.eval_points_tsdf(pi, scene.tsdf_volume, dev)
                    raw, w = rend.eval_points(pi, dec, scene.tsdf_volume, tb, scene.c, stage, dev)
                    res.append(raw[:, 3] if stage == 'high' else raw)
            return torch.cat(res)

        def query_one_call():
            with torch.no_grad():
                return rend.eval_points(pts, dec, scene.tsdf_volume, tb, scene.c, stage, dev)
        t_q, vals = _wall(query, 3, dev, warm=1)
        t_1, (raw1, w1) = _wall(query_one_call, 3, dev, warm=1)
        if band is None:
            band = float((w1 != 1).float().mean())
        macs = MAC_LOW + band * (MAC_HIGH + MAC_ATT) + (MAC_COLOR if stage == 'color' else 0)
        flop = 2.0 * macs * P
        # parity + CPU baseline on a bounded sample: the oracle on 200 000 of the same points
        idx = torch.arange(0, P, max(1, P // 200000), device=dev)[:200000]
        ps = pts[idx].cpu()
        c_cpu = {k: v.cpu() for k, v in scene.c.items()}
        tsdf_cpu = scene.tsdf_volume.cpu()

        def oracle():
            with torch.no_grad():
                return O.eval_points(sd, ps, c_cpu, tsdf_cpu, scene.tsdf_bnds, scene.bound, stage)
        cores = _threads_for_oracle(oracle)
        t_o = min(_cpu_time(oracle) for _ in range(2))
        oraw, ow = oracle()
        got = raw1[idx].cpu()
        sel = slice(3, 4) if stage == 'high' else slice(0, 4)
        fin = oraw[:, 3] != 100
        out['by_stage'][stage] = {
            'ms': t_q * 1e3, 'points_per_s': P / t_q, 'ms_single_call': t_1 * 1e3, 'points_per_s_single_call': P / t_1,
            'in_band_fraction': band, 'algorithmic_tflops': flop / t_1 / 1e12, 'frac_of_f16_mfma_peak_algorithmic': flop / t_1 / 1e12 / PEAK_F16_MFMA_TFLOPS,
            'frac_of_f32_mfma_peak_algorithmic': flop / t_1 / 1e12 / PEAK_F32_MFMA_TFLOPS,
            'cpu_baseline': {'points_per_s': len(idx) / t_o, 'cores': cores, 'kind': 'port', 'sample': f'{len(idx)} evenly strided lattice points ({t_o:.2f} s)'},
            'parity_vs_oracle': {'max_rel': float((got[:, sel] - oraw[:, sel]).abs().max() / oraw[:, sel][fin].abs().max().clamp_min(1e-30)),
                                 'out_of_bound_sets_equal': bool(torch.equal(got[:, 3] == 100, oraw[:, 3] == 100)), 'points': len(idx)}}
    return out


# ----------------------------------------------------------------------------------------------------------------------
def fusion_leg(synthetic, scene, dev):
    """TSDF fusion of one 640x480 RGB-D frame into the room0-sized volume (reference
    src/fusion.py:226-251 launches its kernel over the whole volume per frame)."""
    import numpy as np
    import torch
    from attentive_dfprior_amd.fusion import TSDFVolume
    vol = TSDFVolume(scene.tsdf_bnds.numpy(), 4.0 / 256, device=str(dev))
    n = int(np.prod(vol._vol_dim))
    frames = []
    for k in range(4):
        c2w = scene.default_c2w(offset=(0.2 * k, -0.1 * k, 0.0), yaw=0.3 + 0.5 * k, pitch=-0.1)
        depth = scene.depth_image(c2w)
        color = (torch.rand((scene.H, scene.W, 3), generator=torch.Generator().manual_seed(k)) * 255).floor().to(dev)
        pose = c2w.double().cpu().numpy().copy()
        pose[:3, 1] *= -1.0
        pose[:3, 2] *= -1.0                                            # OpenGL -> OpenCV camera, get_tsdf.py:79-80
        K = np.array([[scene.fx, 0, scene.cx], [0, scene.fy, scene.cy], [0, 0, 1]], dtype=np.float64)
        frames.append((color, depth, K, pose))
    for fr in frames[:1]:
        vol.integrate(*fr)
    torch.cuda.synchronize(dev)
    ts = []
    for fr in frames[1:]:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        vol.integrate(*fr)
        e1.record()
        torch.cuda.synchronize(dev)
        ts.append(e0.elapsed_time(e1) * 1e-3)
    t = sorted(ts)[len(ts) // 2]
    touched = float((vol._weight > 0).float().mean())
    return {'workload': f'one 640x480 RGB-D frame into the room0 volume ({tuple(int(v) for v in vol._vol_dim)} = {n} voxels, tsdf + weight + '
                        'packed colour), whole-volume sweep like the reference', 'ms_per_frame': t * 1e3, 'voxels_per_s': n / t,
            'fraction_of_voxels_observed_after_4_frames': touched,
            'algorithmic_gbps': 24.0 * touched * n / t / 1e9,
            'note': 'algorithmic bytes = 3 volumes x (4 B read + 4 B written) per observed voxel; unobserved quads are rejected on arithmetic alone'}


# ----------------------------------------------------------------------------------------------------------------------
def replica_native_leg(A, synthetic, scene, dec, dev, reps=5):
    """Renderer.render_img at Replica's native camera (module docstring); same scene content (room0 box room, bench grids), same
    decoders, layout caches cleared every frame like the headline step."""
    import torch
    H, W = 680, 1200
    import copy
    sc = copy.copy(scene)                                                # the bench scene's grids and volume behind another camera
    sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy = H, W, 600.0, 600.0, 599.5, 339.5
    rend = A.Renderer(_cfg(32, 16), None, sc)
    tb = sc.tsdf_bnds.to(dev)
    c2w = sc.default_c2w(yaw=0.7, pitch=-0.15)
    gd = sc.depth_image(c2w)

    def frame():
        rend._engine._grid_cache.clear()
        dec._packed.clear()
        return rend.render_img(sc.c, dec, c2w, dev, sc.tsdf_volume, tb, 'color', gt_depth=gd)
    t, out = _wall(frame, reps, dev)
    assert bool(torch.isfinite(out[0]).all()) and bool(torch.isfinite(out[2]).all())
    n = H * W
    return {'ms_per_frame': t * 1e3, 'rays_per_s': n / t, 'rays': n, 'samples_per_ray': 48, 'points': n * 48,
            'far_clamp_segments': (n + rend.ray_batch_size - 1) // rend.ray_batch_size,
            'workload': '1200x680, fx = fy = 600, N_samples 32 + N_surface 16, ray_batch_size 100000, stage color, render_img as one call',
            'reference': 'configs/Replica/replica.yaml:46-53, configs/df_prior.yaml:94-95, src/utils/Renderer.py:278-327'}


def allreduce_model(scene, n_params=15899 + 33410):
    """ring all-reduce of B bytes over N GPUs moves 2 (N-1)/N B per GPU through one xGMI link direction
    (point-to-point links, the ring uses one link per neighbour): t = 2 (N-1)/N B / 153 GB/s + 2 (N-1) hops x ~5 us launch / link
    latency.  A PREDICTION, stated so that the first measured SCALE record can be read against it."""
    grids = sum(int(v.numel()) for v in scene.c.values())
    dense = 4 * (grids + n_params)
    out = {'model': 't = 2 (N-1)/N x bytes / 153 GB/s (one xGMI link per ring neighbour) + 2 (N-1) x 5 us', 'dense_bucket_bytes': dense, 'predicted_ms': {}}
    for n in (2, 4, 8):
        out['predicted_ms'][str(n)] = 2 * (n - 1) / n * dense / (XGMI_LINK_GBPS * 1e9) * 1e3 + 2 * (n - 1) * 5e-3
    return out
