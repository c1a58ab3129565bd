import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


class Mini(object):
    """The committed mini scene (tests/golden/mini_inputs.npz) as CPU tensors."""

    def __init__(self):
        z = np.load(os.path.join(GOLDEN, 'mini_inputs.npz'))
        self.bound = torch.from_numpy(z['bound'])
        self.tsdf_bnds = torch.from_numpy(z['tsdf_bnds'])
        phys = torch.from_numpy(z['tsdf_xyz'])                              # [X,Y,Z] contiguous
        X, Y, Z = phys.shape
        # the reference's permuted, non-contiguous view (get_tsdf.py:95-97)
        self.tsdf_volume = phys.reshape(1, 1, X, Y, Z).permute(0, 1, 4, 3, 2)
        self.c = {k: torch.from_numpy(z[k]) for k in ('grid_low', 'grid_high', 'grid_color')}
        self.sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd.')}
        self.rays_o = torch.from_numpy(z['rays_o'])
        self.rays_d = torch.from_numpy(z['rays_d'])
        self.gt_depth = torch.from_numpy(z['gt_depth'])
        self.gt_color = torch.from_numpy(z['gt_color'])
        self.query_points = torch.from_numpy(z['query_points'])
        self.c2w = torch.from_numpy(z['c2w'])
        self.depth_img = torch.from_numpy(z['depth_img'])
        H, W, fx, fy, cx, cy = z['intrinsics'].tolist()
        self.H, self.W, self.fx, self.fy, self.cx, self.cy = int(H), int(W), fx, fy, cx, cy
        self.n_samples = int(z['n_samples'])
        self.n_surface = int(z['n_surface'])
        self.vol_bnds = self.tsdf_bnds

    def golden(self, name):
        z = np.load(os.path.join(GOLDEN, f'mini_{name}.npz'))
        return {k: z[k] for k in z.files}


@pytest.fixture(scope='session')
def mini():
    return Mini()


def make_cfg(n_samples=32, n_surface=16, lindisp=False, perturb=0.0):
    return {'rendering': {'lindisp': lindisp, 'perturb': perturb, 'N_samples': n_samples,
                          'N_surface': n_surface, 'N_importance': 0},
            'scale': 1, 'occupancy': True, 'meshing': {'resolution': 256}}


def to_dev(x, dev):
    if isinstance(x, dict):
        return {k: to_dev(v, dev) for k, v in x.items()}
    return x.to(dev)


def rel_err(a, b):
    """max |a-b| / max|b|  (global relative error)."""
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


PARITY_FLOOR = 1e-2        # an element smaller than this fraction of its tensor's scale is held to tol x (this fraction x scale)


def assert_close(a, b, tol, what):
    """The parity bar of BASELINE.json's north_star: <= 1e-4 RELATIVE (fp32), element by element:
        |a - b| <= tol * max(|b|, PARITY_FLOOR * max|b|)
    -- purely relative down to 1 % of the tensor's scale; below that the element is held to the absolute error a 1 %-of-scale
    element would be allowed (values crossing zero cannot be held to a ratio).  Rounds 1-4 used tol * (|b| + 0.1 max|b|), which
    let a full-scale element reach 1.1e-4 and gave small elements ten times the room they have now.
    ADFP_PARITY_STATS=<file>: every comparison appends its worst element (as a fraction of its limit) to the file."""
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, f'{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}'
    if a.numel() == 0:
        return
    fin = torch.isfinite(b)
    scale = b[fin].abs().max().clamp_min(1e-30) if fin.any() else torch.tensor(1.0, dtype=torch.float64)
    err = torch.where(fin, (a - b).abs(), torch.zeros_like(b))
    lim = tol * torch.maximum(b.abs(), PARITY_FLOOR * scale)
    both_nan = torch.isnan(a) & torch.isnan(b)
    same_special = both_nan | (~fin & (a == b))                                   # NaN where the reference has NaN, the same infinity
    ratio = torch.where(fin, err / lim, torch.where(same_special, torch.zeros_like(err), torch.full_like(err, float('inf'))))
    ratio = torch.where(torch.isnan(ratio), torch.full_like(ratio, float('inf')), ratio)      # NaN on our side only
    worst = float(ratio.max())
    log = os.environ.get('ADFP_PARITY_STATS')
    if log:
        k = int(ratio.reshape(-1).argmax())
        with open(log, 'a') as f:
            f.write(f'{os.environ.get("ADFP_MATH", "f16x3")} {what} | tol {tol:g} worst {worst:.3f} of the limit; |diff| {float(err.reshape(-1)[k]):.3e} '
                    f'at |ref| {float(b.abs().reshape(-1)[k]):.3e}, scale {float(scale):.3e}, max |diff|/scale {float(err.max() / scale):.2e}\n')
    bad = ratio > 1.0
    assert not bad.any(), (f'{what}: {int(bad.sum())}/{bad.numel()} elements beyond tol={tol} (relative, floor {PARITY_FLOOR} x scale); '
                           f'worst {worst:.2f} x its limit, max abs diff {float(err.max()):.3e}, scale {scale.item():.3e}')


def assert_close_scale(a, b, tol, what, flip_frac=0.0, flip_tol=2e-3):
    """|a-b| <= tol * max|b|: for sums over ~1e5 points (gradients), whose small elements carry the summation-order noise of the
    large ones (float atomics, a different reduction tree than torch's).

    flip_frac > 0 (the f16-split backward, ADFP_MATH=f16x3): that backward takes its ReLU masks from the f16-split forward,
    whose pre-activations differ from torch's by ~1e-6; a unit that lies that close to zero takes the other branch, and that
    ONE sample's contribution appears in / vanishes from the unit's row of the weight gradient (measured against the exact
    backward: every other row agrees to 3e-7, tools/diag_bwd.py) and, through W^T, nudges the earlier layers' rows.  Up to
    `flip_frac` of a tensor's elements may then deviate by up to `flip_tol` x scale; a layout or indexing bug is off by
    O(1) x scale and still fails.  The exact mode (ADFP_MATH=f32) is held to `tol` on every element."""
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, f'{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}'
    scale = b.abs().max().clamp_min(1e-30)
    err = (a - b).abs()
    bad = err > tol * scale + 1e-9
    frac = float(bad.double().mean())
    assert frac <= flip_frac, f'{what}: {frac:.2e} of the elements differ by more than {tol} x scale {scale.item():.3e} (max {err.max().item():.3e})'
    assert err.max() <= flip_tol * scale + 1e-9, f'{what}: max abs diff {err.max().item():.3e} > {flip_tol} x scale {scale.item():.3e}'


# Parameter gradients against torch's autograd.  What separates the two sides is (i) summation order, ~1e-6 of a tensor's
# scale, and (ii) ReLU-boundary samples: a unit whose pre-activation lies within ~1e-6 of zero takes the other branch of relu on
# one side; that ONE sample's contribution then appears in / vanishes from the unit's own row of its layer's weight gradient
# and, because the flipped unit feeds every unit of the layers before it through W^T, shifts ALL rows of the earlier layers (and
# all columns of embedder._B) by that sample's share.  Which samples sit on a boundary changes with every build that moves the
# forward by 1e-7, so the NUMBER of affected rows is not a stable quantity (high_decoder.pts_linears.0.weight of the second-seed
# case: 6 / 15 / 17 of 32 rows beyond 2e-4 x scale in three builds) -- their SIZE is.
# Measured on the MI355X over every parameter tensor of every gradient test (profiles/r03_grad_stats.txt: ~500 comparisons in
# exact-f32 mode, ~900 in f16x3 mode): a handful of tensors have ANY element beyond 2e-4 x scale; the worst element is 5.8e-4 x
# scale, the worst relative Frobenius error 6.2e-4 (high_decoder.embedder._B of the second-seed case, scale 0.95: one boundary
# sample).  The limits are those two figures with < 2x margin, on EVERY element -- the previous criterion let 25 % of the elements
# reach 2e-3 x scale.  A pair of exchanged rows (the negative control in tests/test_gpu_grad.py) is off by 0.89 x scale,
# Frobenius 0.62: three orders of magnitude beyond either limit.  rows_bad is logged (ADFP_GRAD_STATS), not asserted.
PARAM_GRAD_LIMITS = {'f32': dict(cap=1e-3, fro=1e-3), 'f16x3': dict(cap=1e-3, fro=1e-3)}

# The PRIMARY gradient criterion (round 4): the ReLU-boundary effect described above is taken OUT of the comparison instead of
# being absorbed by a tolerance.  The backward exports the ReLU decisions it differentiated along (the f16-split backward: the
# masks its training forward left; the exact backward: what it recomputed, adfp_train_state.dbg_masks_*), the oracle's autograd
# runs with those decisions forced (oracle.adfp_oracle._relu) -- both sides differentiate the SAME piecewise-linear function --
# and EVERY element of every grid and parameter gradient is held to HALF the north-star tolerance (1e-4) in both math modes.
# The looser limits above remain for the comparisons with the reference's own (unforced) autograd gradients in
# tests/golden/mini_<stage>.npz.
TIGHT_GRAD_TOL = {'f32': 5e-5, 'f16x3': 5e-5}      # measured worst element over ~1 150 comparisons: 3.9e-6 x scale, both modes (profiles/r04_grad_stats.txt)


class ReluCapture(object):
    """`cap = ReluCapture(renderer)` before a training call: the backward exports its ReLU decisions (Engine.export_relu_masks)
    and `cap.masks(stage)` returns them on the CPU in the form oracle.adfp_oracle.render_batch_ray(relu_masks=...) takes."""

    def __init__(self, rend):
        self.eng = eng = rend._engine
        eng.export_relu_masks = True
        self.saved = None
        orig = eng.render_backward

        def capturing(decoders, c, tsdf_volume, tsdf_bnds, bound, stage, saved, *a, **k):
            self.saved = saved
            return orig(decoders, c, tsdf_volume, tsdf_bnds, bound, stage, saved, *a, **k)
        eng.render_backward = capturing

    def masks(self, stage):
        def cpu(x):
            if isinstance(x, dict):
                return {k: cpu(v) for k, v in x.items()}
            if isinstance(x, (list, tuple)):
                return [cpu(v) for v in x]
            return x.cpu()
        return cpu(self.eng.relu_masks(self.saved, stage))


def assert_forced_decisions_are_boundary_units(flips):
    """oracle.adfp_oracle.RELU_FLIPS after a forced run: a decision that differs from relu's own is rare (a handful per million
    units) and sits on a pre-activation that the two forwards' rounding (~1e-6 of activations of O(1..10)) can move across
    zero -- forced masks cannot hide a wrong kernel."""
    assert flips['units'] > 0
    assert flips['flipped'] <= 1e-4 * flips['units'] + 2, flips
    assert flips['max_abs_preactivation'] <= 5e-5, flips


def assert_grad_tight(got, ref, what, mode=None):
    mode = mode or os.environ.get('ADFP_MATH', 'f16x3')
    tol = TIGHT_GRAD_TOL[mode]
    a = torch.as_tensor(got).detach().double().cpu()
    b = torch.as_tensor(ref).detach().double().cpu()
    assert a.shape == b.shape, f'{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}'
    scale = b.abs().max().clamp_min(1e-30)
    err = float((a - b).abs().max() / scale)
    log = os.environ.get('ADFP_GRAD_STATS')
    if log:
        with open(log, 'a') as f:
            f.write(f'tight {mode} {what} scale {float(scale):.3e} max {err:.2e}\n')
    assert err <= tol, f'{what} [{mode}, forced ReLU decisions]: max |diff| {err:.2e} x scale {float(scale):.3e} > {tol}'


def param_grad_stats(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    scale = float(b.abs().max().clamp_min(1e-30))
    err = (a - b).abs()
    if a.dim() == 2 and a.shape[0] <= 4:                 # embedder._B [3, 93]: a "row" is a feature column
        rowmax = err.max(dim=0)[0]
    elif a.dim() >= 2:
        rowmax = err.reshape(a.shape[0], -1).max(dim=1)[0]
    else:
        rowmax = err.reshape(-1)
    return dict(scale=scale, max=float(err.max()) / scale, fro=float(err.norm() / b.norm().clamp_min(1e-30)),
                rows_bad=int((rowmax > 2e-4 * scale + 1e-9).sum()), rows=int(rowmax.numel()))


def assert_param_grad_close(got, ref, what, mode=None):
    mode = mode or os.environ.get('ADFP_MATH', 'f16x3')
    a = torch.as_tensor(got).detach()
    b = torch.as_tensor(ref).detach()
    assert a.shape == b.shape, f'{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}'
    st = param_grad_stats(a, b)
    log = os.environ.get('ADFP_GRAD_STATS')
    if log:
        with open(log, 'a') as f:
            f.write(f'{mode} {what} scale {st["scale"]:.3e} max {st["max"]:.2e} fro {st["fro"]:.2e} rows_bad {st["rows_bad"]}/{st["rows"]}\n')
    lim = PARAM_GRAD_LIMITS[mode]
    assert st['max'] <= lim['cap'], f'{what} [{mode}]: max |diff| {st["max"]:.2e} x scale > {lim["cap"]}'
    assert st['fro'] <= lim['fro'], f'{what} [{mode}]: relative Frobenius error {st["fro"]:.2e} > {lim["fro"]}'


def assert_adam_trajectory(a, b, lr, steps, what, tol=2e-4, max_outliers=2e-3):
    """Parameters after a few Adam steps along two paths whose gradients agree to ~1e-5 of their scale: Adam normalises every
    gradient element, so an element whose gradient is noise-sized moves by a full +-lr per step in a direction the noise
    decides.  Almost every element must agree to `tol`; the few outliers may differ by at most the steps themselves."""
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    scale = b.abs().max().clamp_min(1e-30)
    diff = (a - b).abs()
    bad = diff > tol * (b.abs() + 0.1 * scale)
    frac = float(bad.double().mean())
    allowed = max(max_outliers, 4.0 / max(a.numel(), 1))          # a small tensor: a handful of elements, not a fraction
    assert frac <= allowed, f'{what}: {frac:.2e} of the elements left the trajectory'
    assert float(diff.max()) <= 2.0 * lr * steps + 1e-6, f'{what}: an element moved {float(diff.max()):.3e}, more than {steps} Adam steps of lr {lr} allow'
