"""Kernel A/B helper: time the per-stage kernels of one 100 000-ray batch (6.4 M points) with the
library named by ADFP_LIB_PATH (default: the in-tree build).  Not part of the driver contract."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import attentive_dfprior_amd as A                                    # noqa: E402
from attentive_dfprior_amd import synthetic, _lib                    # noqa: E402
from attentive_dfprior_amd.common import get_rays                    # noqa: E402


def main(reps=int(os.environ.get("AB_REPS", "8"))):
    dev = torch.device('cuda:0')
    L = _lib.lib()
    scene = synthetic.Scene('room0', device=dev, grid_std_scale=20.0)
    scene.c['grid_high'] = scene.c['grid_high'] * 100
    dec = A.DF()
    dec.load_state_dict(synthetic.seeded_state_dict(0))
    dec.bound = scene.bound
    dec = dec.to(dev)
    cfg = {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': 48, 'N_surface': 16, 'N_importance': 0},
           'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}
    rend = A.Renderer(cfg, None, scene)
    eng = rend._engine
    tsdf_bnds = scene.tsdf_bnds.to(dev)
    c2w = scene.default_c2w()
    gd_img = scene.depth_image(c2w)
    ro, rd = get_rays(scene.H, scene.W, scene.fx, scene.fy, scene.cx, scene.cy, c2w, dev)
    N, S = int(os.environ.get('AB_RAYS', '100000')), 64
    ro, rd, gd = ro.reshape(-1, 3)[:N].contiguous(), rd.reshape(-1, 3)[:N].contiguous(), gd_img.reshape(-1)[:N].contiguous()
    with torch.no_grad():
        d, u, c, w, aux = eng.render_forward(dec, scene.c, ro, rd, gd, scene.tsdf_volume, tsdf_bnds, scene.bound,
                                             'color', 48, 16, want_aux=True)
    sc, keep = eng.scene(dec, scene.c, scene.tsdf_volume, tsdf_bnds, scene.bound, 'color', images='hg')
    P = N * S
    ap = _lib.AdfpPoints()
    ap.mode, ap.n_points = _lib.PTS_RAYS, P
    ap.rays_o, ap.rays_d, ap.z_vals, ap.S = ro.data_ptr(), rd.data_ptr(), aux['z_vals'].data_ptr(), S
    raw = torch.empty((P, 4), dtype=torch.float32, device=dev)
    wb = torch.empty((P,), dtype=torch.float32, device=dev)
    flags = torch.empty((P,), dtype=torch.uint8, device=dev)
    lst = torch.empty((P,), dtype=torch.int32, device=dev)
    attu = torch.empty((P,), dtype=torch.float32, device=dev)
    cnt = torch.zeros((4,), dtype=torch.int32, device=dev)
    st = _lib.current_stream(dev)

    def timed(fn):
        fn(); fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        return ts[len(ts) // 2], ts[0]
    import hashlib
    sha = hashlib.sha256()
    for t_ in (d, u, c, w, aux['raw'], aux['z_vals']):
        sha.update(t_.contiguous().cpu().numpy().tobytes())
    res = {'lib': os.path.basename(_lib.LIB_PATH), 'checksum': [d.double().sum().item(), u.double().sum().item(), c.double().sum().item()],
           'sha256_of_outputs_raw_z': sha.hexdigest()[:16]}
    res['color_ms'] = timed(lambda: L.adfp_decode_stage(C.byref(sc), C.byref(ap), 2, _lib.ptr(raw), _lib.ptr(wb), _lib.ptr(cnt[1:]), st))
    res['low_ms'] = timed(lambda: L.adfp_decode_stage(C.byref(sc), C.byref(ap), 0, _lib.ptr(raw), _lib.ptr(wb), _lib.ptr(cnt[1:]), st))
    if L.adfp_decode_stage(C.byref(sc), C.byref(ap), 3, _lib.ptr(raw), _lib.ptr(wb), _lib.ptr(cnt[1:]), st) == 0:      # the fused low + colour launch
        res['low_color_ms'] = timed(lambda: L.adfp_decode_stage(C.byref(sc), C.byref(ap), 3, _lib.ptr(raw), _lib.ptr(wb), _lib.ptr(cnt[1:]), st))
    res['tsdf_ms'] = timed(lambda: L.adfp_tsdf_stage(C.byref(sc), C.byref(ap), _lib.ptr(flags), _lib.ptr(lst), _lib.ptr(attu),
                                                     None, _lib.ptr(cnt), st))
    res['batch_ms'] = timed(lambda: eng.render_forward(dec, scene.c, ro, rd, gd, scene.tsdf_volume, tsdf_bnds, scene.bound, 'color', 48, 16))
    res['color_tflops'] = 2.0 * 15575 * P / (res['color_ms'][0] * 1e-3) / 1e12
    res['tsdf_gbps'] = 32.0 * P / (res['tsdf_ms'][0] * 1e-3) / 1e9
    res['in_band_points'] = int(cnt[0].item())
    print(json.dumps(res))
    if hasattr(L, 'adfp_debug_stamps') or os.environ.get('ADFP_STAMPS'):
        # debug build (-DADFP_STAMPS): per-wave start/end wall clock of the last k_decode_h launch
        import numpy as np
        L.adfp_decode_stage(C.byref(sc), C.byref(ap), 2, _lib.ptr(raw), _lib.ptr(wb), _lib.ptr(cnt[1:]), st)
        torch.cuda.synchronize()
        nw = 256 * 12
        buf = (C.c_ulonglong * (2 * nw))()
        L.adfp_debug_stamps.argtypes = [C.c_void_p, C.c_int]
        assert L.adfp_debug_stamps(buf, nw) == 0
        a = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 2).astype(np.int64)
        t0 = a[:, 0].min()
        s0, e0 = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0          # us
        print('wave start us: min %.1f p50 %.1f max %.1f' % (s0.min(), np.median(s0), s0.max()))
        print('wave end   us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f mean %.1f' % (
            e0.min(), np.percentile(e0, 10), np.median(e0), np.percentile(e0, 90), e0.max(), e0.mean()))
        wg_end = e0.reshape(256, 12).max(1)
        print('per-workgroup end us (sorted, every 16th):', np.sort(wg_end)[::16].round(1).tolist())
        print('within-WG spread us (max-min of wave ends): mean %.1f max %.1f' % (
            (e0.reshape(256, 12).max(1) - e0.reshape(256, 12).min(1)).mean(), (e0.reshape(256, 12).max(1) - e0.reshape(256, 12).min(1)).max()))
        x = wg_end.reshape(32, 8)      # blockIdx % 8 = XCD
        print('per-XCD mean WG end us:', x.mean(0).round(1).tolist())
        ph = (C.c_ulonglong * 8)()
        L.adfp_debug_phases.argtypes = [C.c_void_p, C.c_int]
        L.adfp_debug_phases(ph, 1)
        L.adfp_decode_stage(C.byref(sc), C.byref(ap), 2, _lib.ptr(raw), _lib.ptr(wb), _lib.ptr(cnt[1:]), st)
        torch.cuda.synchronize()
        L.adfp_debug_phases(ph, 1)
        tot = float(sum(ph[:5]))
        names = ['ticket/loop', 'point+gather+split c', 'Fourier', '5 layers', 'output+store']
        ntile = P / 32
        print('phase shares of wave-cycles (colour decoder):', {n: round(ph[k] / tot, 3) for k, n in enumerate(names)},
              'wave-cycles per tile:', {n: round(ph[k] / ntile) for k, n in enumerate(names)})


if __name__ == '__main__':
    main()
