"""Gate measurement for weight-gated colour decoding: how many (ray, sample) pairs of a frame have a
compositing weight so small that their colour cannot move any output?

`rgb` enters the outputs only as sum_k w_k c_k with w_k = alpha_k T_k, T_k = prod_{j<k} (1 - alpha_j + 1e-10)
(reference src/common.py:234-247).  T_k never grows along a ray, so "T_k < 2^-30" selects a SUFFIX of every ray: a colour decoder that
skipped those samples would change a colour by at most 2^-30 x |c|.  Reported per frame:

  frac_weight_lt   fraction of samples with w_k < 2^-30 (the verdict's wording)
  frac_T_lt        fraction with T_k < 2^-30 (the gate a kernel can apply: a per-ray prefix length)
  frac_T_lt_16     the same with the prefix rounded UP to 16 samples (what a 16-row MFMA block could skip)
  frac_T_lt_32     ... to 32 samples (one 32-point tile of k_decode_lc16 = half a ray)

on (i) the bench frame (room0, seed-0 decoders, grids at the bench's x20 scale and at the reference's init) and (ii) the office0
map after config 3's 2 640 fused mapping iterations (a training pose and a held-out pose).

  python tools/weight_histogram.py > profiles/r06_weight_histogram.txt
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import attentive_dfprior_amd as A                                    # noqa: E402
from attentive_dfprior_amd import synthetic                          # noqa: E402
from attentive_dfprior_amd.common import get_rays                    # noqa: E402

THR = 2.0 ** -30
CFG = {'rendering': {'lindisp': False, 'perturb': 0.0, 'N_samples': 48, 'N_surface': 16, 'N_importance': 0},
       'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}


def frame_stats(rend, dec, c, sc, tsdf_bnds, c2w, dev, label):
    gd = sc.depth_image(c2w).reshape(-1)
    ro, rd = get_rays(sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, c2w, dev)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    n = ro.shape[0]
    tot = cnt_w = cnt_T = cnt_16 = cnt_32 = 0
    first_half_only = 0
    for i in range(0, n, 100000):
        sl = slice(i, i + 100000)
        with torch.no_grad():
            d, u, col, w, aux = rend._engine.render_forward(dec, c, ro[sl], rd[sl], gd[sl], sc.tsdf_volume, tsdf_bnds, sc.bound, 'color',
                                                            48, 16, want_aux=True)
        raw = aux['raw']
        alpha = torch.sigmoid(10.0 * raw[..., 3].double())
        T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], 1), 1)[:, :-1]
        wgt = alpha * T
        S = alpha.shape[1]
        live = (T >= THR)                                      # a prefix of every ray
        k0 = live.sum(1)                                       # samples whose colour is needed
        tot += alpha.numel()
        cnt_w += int((wgt < THR).sum())
        cnt_T += int((~live).sum())
        cnt_16 += int((S - ((k0 + 15) // 16 * 16).clamp(max=S)).sum())
        cnt_32 += int((S - ((k0 + 31) // 32 * 32).clamp(max=S)).sum())
        first_half_only += int((k0 <= 32).sum())
    print(f'{label}: {n} rays x {S} samples; frac_weight_lt {cnt_w / tot:.4f}  frac_T_lt {cnt_T / tot:.4f}  frac_T_lt_16 {cnt_16 / tot:.4f}  '
          f'frac_T_lt_32 {cnt_32 / tot:.4f}  rays needing <= 32 samples {first_half_only / n:.4f}')
    return cnt_T / tot


def main():
    dev = torch.device('cuda:0')
    print(f'threshold 2^-30 = {THR:.3e}; T_k = prod_(j<k) (1 - alpha_j + 1e-10), alpha = sigmoid(10 occ) (src/common.py:234-239)')
    # (i) the bench frame
    for scale, extra, tag in ((20.0, 100.0, 'bench frame, room0, grids x20 (high x2000)'), (1.0, 1.0, 'bench frame, room0, grids at the reference init')):
        sc = synthetic.Scene('room0', H=480, W=640, device=dev, grid_std_scale=scale)
        sc.c['grid_high'] = sc.c['grid_high'] * extra
        dec = A.DF()
        dec.load_state_dict(synthetic.seeded_state_dict(0))
        dec.bound = sc.bound
        dec = dec.to(dev)
        rend = A.Renderer(CFG, None, sc)
        c2w = sc.default_c2w(yaw=0.3, pitch=-0.1)
        frame_stats(rend, dec, sc.c, sc, sc.tsdf_bnds.to(dev), c2w, dev, tag)
        del sc, dec, rend
    # (ii) config 3's trained state
    import mapping_loop as ML
    run = ML.MappingRun('office0', rays=5000, total_frames=200, fused=True, device=str(dev))
    for f in range(0, 200, 5):
        run.map_frame(f, 300 if f == 0 else 60, ML.LR_FIRST_FACTOR if f == 0 else 1.0)
    torch.cuda.synchronize(dev)
    print(f'office0 after {run.n_iter} fused mapping iterations (config 3):')
    fr = []
    for f, tag in ((100, 'training pose (frame 100)'), (102.5, 'held-out pose (between frames 100 and 105)'), (195, 'training pose (frame 195)')):
        fr.append(frame_stats(run.rend, run.dec, run.c, run.sc, run.tsdf_bnds, ML.circle_pose(run.sc, f, 200), dev, '  ' + tag))
    gate = sum(fr) / len(fr)
    print(f'GATE (VERDICT round 5 item 7: build only if >= 0.40 on the trained state): mean frac_T_lt on the trained state = {gate:.4f} -> '
          + ('BUILD' if gate >= 0.40 else 'do NOT build'))


if __name__ == '__main__':
    main()
