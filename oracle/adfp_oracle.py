"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product.

CPU restatement (plain PyTorch ops on CPU tensors) of the per-ray volume-rendering
hot path of MachinePerceptionLab/Attentive_DFPrior.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker / the timed CPU baseline.  The product package
``attentive_dfprior_amd`` never imports it and has no CPU fallback.

Parity pinning: the reference holds no tests or golden vectors for this path
(SURVEY.md section 4), so this restatement is pinned against the reference itself,
imported read-only in the build container by ``oracle/validate_against_reference.py``
(0.0 max-abs difference on forward) and against the committed
fixtures under ``tests/golden/`` that ``tests/golden/make_golden.py`` generated from
the reference's own modules.

Mapper-side restatements (src.Mapper needs cv2 / colorama and cannot be imported in the build container):
``prefilter_mask`` (src/Mapper.py:440-445) and ``frustum_mask_np`` (src/Mapper.py:90-158) are pinned by vectors that
``tests/golden/make_mapper_golden.py`` produced by EXECUTING the reference's own source lines (the ten inline lines
of optimize_map; get_mask_from_c2w as a method of a stub object) -- ``tests/test_oracle_golden.py`` checks both
exactly.  In get_mask_from_c2w one name is substituted: ``cv2.remap`` (opencv-python==4.5.5.64,
environment.yaml:194, absent here) resolves to ``remap_linear_np`` below, a restatement of OpenCV's documented
bilinear remap; that ONE function stays PARITY UNPINNED and says so at its definition.  ``tsdf_integrate_np`` (the CUDA
kernel string of src/fusion.py:69-142) is pinned since round 4: ``oracle/build_ref_fusion.py`` compiles the reference's own
kernel string with hipcc from where it lies and ``tests/test_gpu_fusion.py`` holds the restatement to it bit for bit on the
MI355X.  Everything on the render path proper (rows a1-a15) is pinned as above.

Every function cites the reference file:line it follows (paths relative to the
reference checkout).  Decoder weights are passed as a flat ``dict`` keyed exactly like
``DF.state_dict()`` in the reference (``low_decoder.fc_c.0.weight`` ...), so the
oracle does not depend on any nn.Module of the product.

dtype conventions follow the reference exactly: rays f32, ``bound``/``tsdf_bnds`` f64,
z_vals/pts f64, normalised coords cast to f32 before ``grid_sample``, decoders f32,
depth/uncertainty f64, colour/weight f32.
"""
import torch
import torch.nn.functional as F

STAGES = ('low', 'high', 'color')
DECODERS = ('low', 'high', 'color')


# ----------------------------------------------------------------------------------
# a1/a2: rays  (src/common.py:254-272, :76-91)
# ----------------------------------------------------------------------------------
def get_rays(H, W, fx, fy, cx, cy, c2w):
    """src/common.py:254-272.  Directions are NOT normalised."""
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H), indexing='ij')
    i = i.t()
    j = j.t()
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)
    dirs = dirs.reshape(H, W, 1, 3)
    rays_d = torch.sum(dirs * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def get_rays_from_uv(i, j, c2w, fx, fy, cx, cy):
    """src/common.py:76-91."""
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)
    dirs = dirs.reshape(-1, 1, 3)
    rays_d = torch.sum(dirs * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def select_uv(H0, H1, W0, W1, indices, depth, color):
    """src/common.py:94-124 with the random ``indices`` supplied by the caller
    (the reference draws them with torch.randint, common.py:101)."""
    depth = depth[H0:H1, W0:W1]
    color = color[H0:H1, W0:W1]
    i, j = torch.meshgrid(torch.linspace(W0, W1 - 1, W1 - W0), torch.linspace(H0, H1 - 1, H1 - H0),
                          indexing='ij')
    i = i.t().reshape(-1)
    j = j.t().reshape(-1)
    return i[indices], j[indices], depth.reshape(-1)[indices], color.reshape(-1, 3)[indices]


# ----------------------------------------------------------------------------------
# a4: sampler  (src/utils/Renderer.py:134-225)
# ----------------------------------------------------------------------------------
def sample_z(rays_o, rays_d, gt_depth, bound, N_samples, N_surface, lindisp=False, perturb=0.0,
             t_rand=None, depth_max=None):
    """z_vals [N, S] (f64 when bound is f64) -- src/utils/Renderer.py:134-221.
    ``t_rand`` replaces the reference's torch.rand draw (Renderer.py:216) when perturb > 0.
    ``depth_max`` (not in the reference) replaces max(gt_depth) of Renderer.py:159/:195 so that a
    ray shard can be sampled exactly like the full batch it was cut from."""
    if gt_depth is None:
        N_surface = 0
        near = 0.01
    else:
        gt_depth = gt_depth.reshape(-1, 1)
        near = gt_depth.repeat(1, N_samples) * 0.01
    det_rays_o = rays_o.detach().unsqueeze(-1)
    det_rays_d = rays_d.detach().unsqueeze(-1)
    t = (bound.unsqueeze(0) - det_rays_o) / det_rays_d                     # Renderer.py:151
    far_bb, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
    far_bb = far_bb.unsqueeze(-1)
    far_bb = far_bb + 0.01
    if gt_depth is not None:
        dmax = torch.max(gt_depth) if depth_max is None else depth_max.reshape(()).to(gt_depth.dtype)
        far = torch.clamp(far_bb, 0, dmax * 1.2)                           # Renderer.py:159
    else:
        far = far_bb
    if N_surface > 0:
        nz = gt_depth > 0                                                  # Renderer.py:179
        gt_nz = gt_depth[nz].unsqueeze(-1).repeat(1, N_surface)
        ts = torch.linspace(0., 1., steps=N_surface, device=rays_o.device).double()
        z_nz = 0.95 * gt_nz * (1. - ts) + 1.05 * gt_nz * ts                # Renderer.py:186
        z_surf = torch.zeros(gt_depth.shape[0], N_surface, device=rays_o.device).double()
        nz = nz.squeeze(-1)
        z_surf[nz, :] = z_nz
        far_surface = dmax
        z_zero = 0.001 * (1. - ts) + far_surface * ts                      # Renderer.py:196
        z_surf[~nz, :] = z_zero
    t_vals = torch.linspace(0., 1., steps=N_samples, device=rays_o.device)
    if not lindisp:
        z_vals = near * (1. - t_vals) + far * t_vals                       # Renderer.py:206
    else:
        z_vals = 1. / (1. / near * (1. - t_vals) + 1. / far * t_vals)
    if perturb > 0.:
        mids = .5 * (z_vals[..., 1:] + z_vals[..., :-1])                   # Renderer.py:212
        upper = torch.cat([mids, z_vals[..., -1:]], -1)
        lower = torch.cat([z_vals[..., :1], mids], -1)
        z_vals = lower + (upper - lower) * t_rand
    if N_surface > 0:
        z_vals, _ = torch.sort(torch.cat([z_vals, z_surf.double()], -1), -1)  # Renderer.py:220
    return z_vals


# ----------------------------------------------------------------------------------
# a3: the Mapper's bounding-box pre-filter  (src/Mapper.py:438-449)
# pinned by tests/golden/mapper_prefilter.npz = the output of those very lines, executed from the reference's source
# ----------------------------------------------------------------------------------
def prefilter_mask(rays_o, rays_d, gt_depth, bound):
    """inside_mask of src/Mapper.py:440-445 (bound f64 [3,2], rays f32)."""
    det_rays_o = rays_o.clone().detach().unsqueeze(-1)                     # (N, 3, 1)
    det_rays_d = rays_d.clone().detach().unsqueeze(-1)
    t = (bound.unsqueeze(0) - det_rays_o) / det_rays_d
    t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
    return t >= gt_depth


# ----------------------------------------------------------------------------------
# a6/a7/a10: normalisation and trilinear lookup  (src/common.py:275-290, decoder.py:168-175)
# ----------------------------------------------------------------------------------
def normalize_3d_coordinate(p, bound):
    """src/common.py:275-290 (out of place)."""
    p = p.reshape(-1, 3)
    x = ((p[:, 0] - bound[0, 0]) / (bound[0, 1] - bound[0, 0])) * 2 - 1.0
    y = ((p[:, 1] - bound[1, 0]) / (bound[1, 1] - bound[1, 0])) * 2 - 1.0
    z = ((p[:, 2] - bound[2, 0]) / (bound[2, 1] - bound[2, 0])) * 2 - 1.0
    return torch.stack([x, y, z], -1)


def trilerp(vol, p, bound):
    """F.grid_sample 5-D, bilinear, border, align_corners=True -- decoder.py:168-175, :295-303.
    vol [1,C,Z,Y,X] (any strides), p [P,3] world coords -> [C,P]."""
    p_nor = normalize_3d_coordinate(p, bound).unsqueeze(0)
    vgrid = p_nor[:, :, None, None].float()
    out = F.grid_sample(vol, vgrid, padding_mode='border', align_corners=True, mode='bilinear')
    return out.squeeze(-1).squeeze(-1).squeeze(0)


def trilerp_explicit(vol, p, bound):
    """The same lookup written out corner by corner (ATen grid_sampler_3d semantics:
    unnormalise ((x+1)/2)*(size-1), clip to [0,size-1], floor, 8 weighted corners with
    out-of-range corners contributing zero).  Used by tests to pin the arithmetic the HIP
    kernels implement."""
    _, C, D, Hh, Ww = vol.shape
    pn = normalize_3d_coordinate(p, bound).float()

    def unnorm(c, size):
        c = ((c + 1.0) / 2.0) * (size - 1)
        return torch.clamp(c, 0.0, float(size - 1))
    ix, iy, iz = unnorm(pn[:, 0], Ww), unnorm(pn[:, 1], Hh), unnorm(pn[:, 2], D)
    x0, y0, z0 = torch.floor(ix), torch.floor(iy), torch.floor(iz)
    tx, ty, tz = ix - x0, iy - y0, iz - z0
    x0, y0, z0 = x0.long(), y0.long(), z0.long()
    out = torch.zeros(C, p.shape[0], device=p.device)
    v = vol[0]
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                w = (tx if dx else 1 - tx) * (ty if dy else 1 - ty) * (tz if dz else 1 - tz)
                xi, yi, zi = x0 + dx, y0 + dy, z0 + dz
                ok = (xi < Ww) & (yi < Hh) & (zi < D)
                val = v[:, zi.clamp(max=D - 1), yi.clamp(max=Hh - 1), xi.clamp(max=Ww - 1)]
                out = out + torch.where(ok, w, torch.zeros_like(w)) * val
    return out


# ----------------------------------------------------------------------------------
# a8/a9: decoder MLP  (decoder.py:26-30, :177-203)
# ----------------------------------------------------------------------------------
def _relu(h, forced=None, only=None):
    """F.relu(h) -- or, test-only, the SAME piecewise-linear branch a kernel took: `forced` bool [..., units] says which units
    pass (`only` bool [...]: rows the forced decisions apply to; the others keep relu's own).  A unit whose pre-activation lies
    within rounding of zero is decided either way by two correct implementations, and its whole weight-gradient row moves with
    the decision; forcing the kernel's decisions makes the oracle differentiate the function the kernels differentiated, so
    that the comparison can be held to the forward tolerance (tests/test_gpu_grad.py)."""
    if forced is None:
        return F.relu(h)
    if only is not None:
        forced = torch.where(only.unsqueeze(-1), forced, h > 0)
    flipped = forced != (h > 0)
    RELU_FLIPS['units'] += forced.numel()
    RELU_FLIPS['flipped'] += int(flipped.sum())
    if flipped.any():
        RELU_FLIPS['max_abs_preactivation'] = max(RELU_FLIPS['max_abs_preactivation'], float(h.detach()[flipped].abs().max()))
    return torch.where(forced, h, torch.zeros_like(h))


# what forcing cost (test-only bookkeeping of _relu): a forced decision that differs from relu's own must sit on a unit whose
# pre-activation is within rounding of zero -- the tests assert it, so that forced masks cannot hide a wrong kernel
RELU_FLIPS = {'units': 0, 'flipped': 0, 'max_abs_preactivation': 0.0}


def reset_relu_flips():
    RELU_FLIPS.update(units=0, flipped=0, max_abs_preactivation=0.0)


def mlp_forward(sd, name, p, c_grid, bound, relu_mask=None, relu_rows=None):
    """MLP.forward, decoder.py:177-203.  sd: state dict, name in DECODERS, p [P,3] f64/f32.
    relu_mask (test-only, see _relu): bool [P, 5, 32]; relu_rows: bool [P] rows it applies to."""
    pre = f'{name}_decoder.'
    c = trilerp(c_grid['grid_' + name], p, bound).t()                       # decoder.py:179-180
    if name == 'high':                                                      # concat_feature, :182-187
        with torch.no_grad():
            c_low = trilerp(c_grid['grid_low'], p, bound).t()
        c = torch.cat([c, c_low], dim=1)
    pf = p.float()
    emb = torch.sin(pf @ sd[pre + 'embedder._B'])                           # decoder.py:29-30
    h = emb
    for i in range(5):
        h = F.linear(h, sd[pre + f'pts_linears.{i}.weight'], sd[pre + f'pts_linears.{i}.bias'])
        h = _relu(h, None if relu_mask is None else relu_mask[:, i], relu_rows)
        h = h + F.linear(c, sd[pre + f'fc_c.{i}.weight'], sd[pre + f'fc_c.{i}.bias'])
        if i == 2:
            h = torch.cat([emb, h], -1)
    out = F.linear(h, sd[pre + 'output_linear.weight'], sd[pre + 'output_linear.bias'])
    if name != 'color':
        out = out.squeeze(-1)
    return out


def inv_tsdf(tsdf_val):
    """decoder.py:244-248."""
    s = 1. - (tsdf_val + 1.) / 2.
    s = torch.clamp(s, 0.0, 1.0)
    u = -0.1 * torch.log((1 / (s + 1e-8)) - 1 + 1e-7)
    return torch.clamp(u, -100.0, 100.0)


def mlp_tsdf_forward(sd, occ, tsdf_val, relu_masks=None):
    """mlp_tsdf.forward, decoder.py:240-258, given the TSDF value at the points.  relu_masks (test-only, see _relu): four bool
    tensors [M, 64 / 128 / 128 / 64]."""
    u = inv_tsdf(tsdf_val)
    inp = torch.stack([occ, u], dim=1)
    h = inp
    for i in range(4):
        h = _relu(F.linear(h, sd[f'mlp.pts_linears.{i}.weight'], sd[f'mlp.pts_linears.{i}.bias']),
                  None if relu_masks is None else relu_masks[i])
    a = torch.softmax(F.linear(h, sd['mlp.output_linear.weight'], sd['mlp.output_linear.bias']), dim=1)
    out = (a * inp).sum(dim=1)
    return out, a[:, 1]


# ----------------------------------------------------------------------------------
# a12: DF.forward  (decoder.py:307-353)
# ----------------------------------------------------------------------------------
def df_forward(sd, p, c_grid, tsdf_volume, tsdf_bnds, bound, stage, return_aux=False, relu_masks=None):
    """p [P,3] -> raw [P,4], w [P].  relu_masks (test-only, see _relu): the dict Engine.relu_masks returns, on p's device."""
    P = p.shape[0]
    rm = relu_masks or {}
    low = mlp_forward(sd, 'low', p, c_grid, bound, rm.get('low'))
    aux = {}
    if stage == 'low':
        raw = torch.zeros(P, 4, device=p.device)
        raw = torch.cat([raw[:, :3], low.unsqueeze(-1)], -1)
        w = torch.ones(P, device=p.device)
        return (raw, w, aux) if return_aux else (raw, w)
    high = mlp_forward(sd, 'high', p, c_grid, bound, rm.get('high'), rm.get('high_valid'))
    if stage == 'color':
        rgb = mlp_forward(sd, 'color', p, c_grid, bound, rm.get('color'))[:, :3]
    else:
        rgb = torch.zeros(P, 3, device=p.device)
    f_add = high + low                                                       # decoder.py:325/:342
    t = trilerp(tsdf_volume, p, tsdf_bnds).reshape(-1)
    mask = (t > -1.0 + 1e-4) & (t < 1.0 - 1e-4)                              # decoder.py:329/:346
    if 'band' in rm and not torch.equal(rm['band'], mask):
        raise AssertionError('relu_masks: the kernels and the oracle disagree on which points are in the TSDF band')
    fused, a1 = mlp_tsdf_forward(sd, f_add[mask], t[mask], [m[mask] for m in rm['att']] if 'att' in rm else None)
    occ = low.clone()
    occ[mask] = fused                                                        # unmasked keep LOW only
    w = torch.ones(P, device=p.device)
    w[mask] = a1
    raw = torch.cat([rgb, occ.unsqueeze(-1)], -1)
    if return_aux:
        aux = {'tsdf': t, 'band': mask}
        return raw, w, aux
    return raw, w


def eval_points(sd, p, c_grid, tsdf_volume, tsdf_bnds, bound, stage, return_aux=False, relu_masks=None):
    """Renderer.eval_points, src/utils/Renderer.py:27-71 (chunking is value-neutral)."""
    mask = ((p[:, 0] < bound[0][1]) & (p[:, 0] > bound[0][0]) &
            (p[:, 1] < bound[1][1]) & (p[:, 1] > bound[1][0]) &
            (p[:, 2] < bound[2][1]) & (p[:, 2] > bound[2][0]))
    res = df_forward(sd, p, c_grid, tsdf_volume, tsdf_bnds, bound, stage, return_aux, relu_masks)
    raw, w = res[0], res[1]
    occ = torch.where(mask, raw[:, 3], torch.full_like(raw[:, 3], 100.0))  # Renderer.py:64
    raw = torch.cat([raw[:, :3], occ.unsqueeze(-1)], -1)
    if return_aux:
        aux = res[2]
        aux['inbound'] = mask
        return raw, w, aux
    return raw, w


# ----------------------------------------------------------------------------------
# a13: compositing  (src/common.py:206-251, occupancy=True branch)
# ----------------------------------------------------------------------------------
def raw2outputs(raw, z_vals):
    rgb = raw[..., :3]
    alpha = torch.sigmoid(10 * raw[..., 3])                                  # common.py:236
    ones = torch.ones((alpha.shape[0], 1), device=alpha.device)
    weights = alpha.float() * torch.cumprod(
        torch.cat([ones, (1. - alpha + 1e-10).float()], -1).float(), -1)[:, :-1]
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z_vals, -1)
    tmp = z_vals - depth_map.unsqueeze(-1)
    depth_var = torch.sum(weights * tmp * tmp, dim=1)
    return depth_map, depth_var, rgb_map, weights


# ----------------------------------------------------------------------------------
# a4..a13: render_batch_ray  (src/utils/Renderer.py:110-255) and a14 render_img (:258-327)
# ----------------------------------------------------------------------------------
def render_batch_ray(sd, c_grid, rays_d, rays_o, tsdf_volume, tsdf_bnds, bound, stage, gt_depth,
                     N_samples, N_surface, lindisp=False, perturb=0.0, t_rand=None, return_aux=False,
                     depth_max=None, relu_masks=None):
    """Returns (depth f64 [N], uncertainty f64 [N], color f32 [N,3], weight f32 [N,S,1])."""
    N = rays_o.shape[0]
    z_vals = sample_z(rays_o, rays_d, gt_depth, bound, N_samples, N_surface, lindisp, perturb, t_rand, depth_max)
    S = z_vals.shape[1]
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]   # Renderer.py:223
    res = eval_points(sd, pts.reshape(-1, 3), c_grid, tsdf_volume, tsdf_bnds, bound, stage, return_aux, relu_masks)
    raw = res[0].reshape(N, S, 4)
    weight = res[1].reshape(N, S, 1)
    depth, var, color, cw = raw2outputs(raw, z_vals)
    if return_aux:
        aux = res[2]
        aux.update({'z_vals': z_vals, 'raw': raw, 'composite_weights': cw})
        return depth, var, color, weight, aux
    return depth, var, color, weight


def render_img(sd, c_grid, c2w, H, W, fx, fy, cx, cy, tsdf_volume, tsdf_bnds, bound, stage, gt_depth,
               N_samples, N_surface, ray_batch_size=100000):
    """src/utils/Renderer.py:258-327.  The far clamp uses the per-batch max depth (:159)."""
    with torch.no_grad():
        rays_o, rays_d = get_rays(H, W, fx, fy, cx, cy, c2w)
        rays_o = rays_o.reshape(-1, 3)
        rays_d = rays_d.reshape(-1, 3)
        gt_depth = gt_depth.reshape(-1)
        ds, us, cs = [], [], []
        for i in range(0, rays_d.shape[0], ray_batch_size):
            d, u, c, _ = render_batch_ray(sd, c_grid, rays_d[i:i + ray_batch_size], rays_o[i:i + ray_batch_size],
                                          tsdf_volume, tsdf_bnds, bound, stage, gt_depth[i:i + ray_batch_size],
                                          N_samples, N_surface)
            ds.append(d.double())
            us.append(u.double())
            cs.append(c)
        return (torch.cat(ds).reshape(H, W), torch.cat(us).reshape(H, W), torch.cat(cs).reshape(H, W, 3))


# ----------------------------------------------------------------------------------
# a15: Mapper loss  (src/Mapper.py:457-469)
# ----------------------------------------------------------------------------------
def mapper_loss(depth, color, weight, gt_depth, gt_color, stage, warmup=False, w_color_loss=0.2):
    depth_mask = gt_depth > 0
    loss = torch.abs(gt_depth[depth_mask] - depth[depth_mask]).sum()
    if warmup:
        loss = loss + torch.abs(weight - torch.ones(weight.shape, device=weight.device)).sum()
    if stage == 'color':
        loss = loss + w_color_loss * torch.abs(gt_color - color).sum()
    return loss


def tracker_loss(depth, uncertainty, color, gt_depth, gt_color, handle_dynamic=True, w_color_loss=0.5):
    """Camera-tracking loss, src/Tracker.py:116-129 (use_color_in_tracking: True)."""
    uncertainty = uncertainty.detach()
    if handle_dynamic:
        tmp = torch.abs(gt_depth - depth) / torch.sqrt(uncertainty + 1e-10)
        mask = (tmp < 10 * tmp.median()) & (gt_depth > 0)
    else:
        mask = gt_depth > 0
    loss = (torch.abs(gt_depth - depth) / torch.sqrt(uncertainty + 1e-10))[mask].sum()
    loss = loss + w_color_loss * torch.abs(gt_color - color)[mask].sum()
    return loss


# ----------------------------------------------------------------------------------
# helpers shared by tests / bench (synthetic scene of SURVEY.md section 8d)
# ----------------------------------------------------------------------------------
def decoder_param_shapes(c_dim=32, hidden=32, emb=93):
    """Parameter names/shapes of DF.state_dict() (decoder.py:110-166, :212-228, :276-292),
    in registration order."""
    shapes = []
    for name, cd, nout in (('low', c_dim, 1), ('high', 2 * c_dim, 1), ('color', c_dim, 4)):
        pre = f'{name}_decoder.'
        for i in range(5):
            shapes += [(pre + f'fc_c.{i}.weight', (hidden, cd)), (pre + f'fc_c.{i}.bias', (hidden,))]
        shapes += [(pre + 'embedder._B', (3, emb))]
        ins = [emb, hidden, hidden, hidden + emb, hidden]
        for i in range(5):
            shapes += [(pre + f'pts_linears.{i}.weight', (hidden, ins[i])),
                       (pre + f'pts_linears.{i}.bias', (hidden,))]
        shapes += [(pre + 'output_linear.weight', (nout, hidden)), (pre + 'output_linear.bias', (nout,))]
    dims = [2, 64, 128, 128, 64]
    for i in range(4):
        shapes += [(f'mlp.pts_linears.{i}.weight', (dims[i + 1], dims[i])),
                   (f'mlp.pts_linears.{i}.bias', (dims[i + 1],))]
    shapes += [('mlp.output_linear.weight', (2, 64)), ('mlp.output_linear.bias', (2,))]
    return shapes


def random_state_dict(seed=0, bias_scale=0.05, occ_bias=-0.5, out_scale=0.15):
    """Seeded weights with the reference's init distributions (xavier-uniform weights,
    N(0,25^2) Fourier matrix, nn.Linear default for fc_c) but non-zero biases so that
    bias handling is exercised.  The occupancy heads are damped (out_scale) and biased
    negative (occ_bias) so that free space is mostly transparent and compositing weights
    spread over many samples, as with the pretrained decoders the snapshot lacks."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in decoder_param_shapes():
        if k.endswith('_B'):
            sd[k] = torch.randn(shp, generator=g) * 25
        elif k.endswith('weight'):
            fan_out, fan_in = shp
            gain = 1.0 if 'output_linear' in k else 2 ** 0.5
            if 'fc_c' in k:
                a = (1.0 / fan_in) ** 0.5
            else:
                a = gain * (6.0 / (fan_in + fan_out)) ** 0.5
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * a
        else:
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * bias_scale
    for name in ('low', 'high'):
        sd[f'{name}_decoder.output_linear.weight'] *= out_scale
    sd['low_decoder.output_linear.bias'] += occ_bias
    sd['color_decoder.output_linear.weight'][3] *= out_scale
    return sd


# ----------------------------------------------------------------------------------
# TSDF fusion: numpy restatement of the reference's CUDA kernel
# src/fusion.py:69-142 in float32, statement by statement.  PINNED (round 4): the reference's kernel string itself is
# compiled with hipcc by oracle/build_ref_fusion.py (neither pycuda nor numba exists in the build container, but the
# string is plain CUDA C) and tests/test_gpu_fusion.py::test_integrate_matches_the_reference_kernel holds this function
# to that build bit for bit (tsdf, weight, packed colour; 13.9 M voxels, three frames) on the MI355X.
# ----------------------------------------------------------------------------------
def tsdf_integrate_np(tsdf, weight, color, origin, voxel, cam_intr, cam_pose, color_im_packed, depth_im, trunc, obs_w):
    import numpy as np
    f = np.float32
    dx, dy, dz = tsdf.shape
    n = dx * dy * dz
    idx = np.arange(n, dtype=np.int32)
    vx = np.floor(idx.astype(f) / f(dy * dz)).astype(f)                                   # fusion.py:92
    vy = np.floor((idx - vx.astype(np.int32) * (dy * dz)).astype(f) / f(dz)).astype(f)       # :93
    vz = (idx - vx.astype(np.int32) * (dy * dz) - vy.astype(np.int32) * dz).astype(f)        # :94
    origin = np.asarray(origin, f); P = np.asarray(cam_pose, f).reshape(4, 4); K = np.asarray(cam_intr, f).reshape(3, 3)
    voxel = f(voxel); trunc = f(trunc); obs_w = f(obs_w)
    px, py, pz = origin[0] + vx * voxel, origin[1] + vy * voxel, origin[2] + vz * voxel
    tx, ty, tz = px - P[0, 3], py - P[1, 3], pz - P[2, 3]
    cx = P[0, 0] * tx + P[1, 0] * ty + P[2, 0] * tz                                          # :104-106
    cy = P[0, 1] * tx + P[1, 1] * ty + P[2, 1] * tz
    cz = P[0, 2] * tx + P[1, 2] * ty + P[2, 2] * tz

    def roundf(x):                                                                            # C roundf: half away from zero
        return np.where(x >= 0, np.floor(x + f(0.5)), np.ceil(x - f(0.5))).astype(f)
    with np.errstate(divide='ignore', invalid='ignore'):
        ux = K[0, 0] * (cx / cz) + K[0, 2]
        uy = K[1, 1] * (cy / cz) + K[1, 2]
    # roundf of x + 0.5 can differ from true roundf when x + 0.5 rounds up; use exact comparison instead
    def c_roundf(x):
        r = np.trunc(x)
        frac = np.abs(x - r)
        return (r + np.sign(x) * (frac >= f(0.5))).astype(f)
    h, w = depth_im.shape
    with np.errstate(invalid='ignore'):
        pix_x = np.nan_to_num(c_roundf(ux), nan=-1e9, posinf=1e9, neginf=-1e9).clip(-2e9, 2e9).astype(np.int64)
        pix_y = np.nan_to_num(c_roundf(uy), nan=-1e9, posinf=1e9, neginf=-1e9).clip(-2e9, 2e9).astype(np.int64)
    ok = (pix_x >= 0) & (pix_x < w) & (pix_y >= 0) & (pix_y < h) & ~(cz < 0)
    dval = np.zeros(n, f)
    dval[ok] = depth_im[pix_y[ok], pix_x[ok]]
    ok &= dval != 0
    diff = dval - cz
    ok &= ~(diff < -trunc)
    dist = np.minimum(f(1.0), diff / trunc)
    t, wt, col = tsdf.reshape(-1).copy(), weight.reshape(-1).copy(), color.reshape(-1).copy()
    w_old = wt[ok]
    w_new = w_old + obs_w
    wt[ok] = w_new
    t[ok] = (t[ok] * w_old + obs_w * dist[ok]) / w_new
    oc = col[ok]
    ob = np.floor(oc / f(65536)); og = np.floor((oc - ob * f(65536)) / f(256)); orr = oc - ob * f(65536) - og * f(256)
    nc = color_im_packed[pix_y[ok], pix_x[ok]].astype(f)
    nb = np.floor(nc / f(65536)); ng = np.floor((nc - nb * f(65536)) / f(256)); nr = nc - nb * f(65536) - ng * f(256)
    nb = np.minimum(c_roundf((ob * w_old + obs_w * nb) / w_new), f(255))
    ng = np.minimum(c_roundf((og * w_old + obs_w * ng) / w_new), f(255))
    nr = np.minimum(c_roundf((orr * w_old + obs_w * nr) / w_new), f(255))
    col[ok] = nb * f(65536) + ng * f(256) + nr
    return t.reshape(tsdf.shape), wt.reshape(tsdf.shape), col.reshape(tsdf.shape)


# ----------------------------------------------------------------------------------
# frustum feature selection  (src/Mapper.py:90-158)
# frustum_mask_np is pinned by tests/golden/mapper_frustum.npz (the reference's get_mask_from_c2w executed from its
# source, tests/golden/make_mapper_golden.py).  PARITY UNPINNED: remap_linear_np only -- cv2 is absent, so
# cv2.remap is restated from OpenCV 4.5.5's documented algorithm (imgproc remap, INTER_LINEAR: map rounded to 1/32
# pixel with cvRound, weights from the 32x32 bilinear table, BORDER_CONSTANT value 0 for taps outside the image).
# ----------------------------------------------------------------------------------
def remap_linear_np(img, u, v):
    import numpy as np
    H, W = img.shape
    su = np.rint(u.astype(np.float32) * np.float32(32)).astype(np.int64)
    sv = np.rint(v.astype(np.float32) * np.float32(32)).astype(np.int64)
    sx = np.clip(su >> 5, -32768, 32767)
    sy = np.clip(sv >> 5, -32768, 32767)
    fx = ((su & 31).astype(np.float32)) * np.float32(1 / 32)
    fy = ((sv & 31).astype(np.float32)) * np.float32(1 / 32)
    one = np.float32(1)
    w = [(one - fy) * (one - fx), (one - fy) * fx, fy * (one - fx), fy * fx]

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        return np.where(ok, img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], np.float32(0)).astype(np.float32)
    out = tap(sy, sx) * w[0] + tap(sy, sx + 1) * w[1] + tap(sy + 1, sx) * w[2] + tap(sy + 1, sx + 1) * w[3]
    gone = (sx >= W) | (sx + 1 < 0) | (sy >= H) | (sy + 1 < 0)
    return np.where(gone, np.float32(0), out).astype(np.float32)


def _frustum_points(val_shape, bound):
    Z, Y, X = val_shape
    gx, gy, gz = torch.meshgrid(torch.linspace(bound[0][0], bound[0][1], X), torch.linspace(bound[1][0], bound[1][1], Y),
                                torch.linspace(bound[2][0], bound[2][1], Z), indexing='ij')      # Mapper.py:105-107
    return torch.stack([gx, gy, gz], -1).reshape(-1, 3)


def _frustum_decide(cam, pts, c2w, depth_np, H, W, fx, fy, cx, cy, dmax=None, dist2=None):
    """Mapper.py:118-153 from the camera-space coordinates on: returns (mask, max of the looked-up depths)."""
    import numpy as np
    K = np.array([[fx, .0, cx], [.0, fy, cy], [.0, .0, 1.0]])
    cam = cam.copy()
    cam[:, 0] *= -1
    uv = K @ cam                                                            # f64 from here, :120
    z = uv[:, -1:] + 1e-5
    uv = (uv[:, :2] / z).astype(np.float32)
    depths = remap_linear_np(depth_np, uv[:, 0, 0], uv[:, 1, 0]).reshape(-1, 1)
    mask = (uv[:, 0] < W) * (uv[:, 0] > 0) * (uv[:, 1] < H) * (uv[:, 1] > 0)
    top = np.max(depths)
    depths[depths == 0] = top if dmax is None else dmax                     # :138-140
    mask = mask & (0 <= -z[:, :, 0]) & (-z[:, :, 0] <= depths + 0.5)
    mask = mask.reshape(-1)
    if dist2 is None:
        dist = pts - torch.from_numpy(c2w[:3, 3]).unsqueeze(0)              # :146-151
        dist2 = torch.sum(dist * dist, 1).numpy()
    mask = mask | (dist2 < 0.5 * 0.5)
    return mask, top


def frustum_mask_np(c2w, val_shape, depth_np, bound, H, W, fx, fy, cx, cy):
    """Mapper.get_mask_from_c2w; returns the bool mask in the grid tensor's [Z, Y, X] order
    (= the reference's [X, Y, Z] result after the permute(2, 1, 0) of src/Mapper.py:345)."""
    import numpy as np
    Z, Y, X = val_shape
    pts = _frustum_points(val_shape, bound)
    c2w = c2w.cpu().numpy()
    w2c = np.linalg.inv(c2w)
    homo = np.concatenate([pts.numpy(), np.ones((pts.shape[0], 1), np.float32)], 1).reshape(-1, 4, 1)
    cam = (w2c @ homo)[:, :3]                                               # :116-117
    mask, _ = _frustum_decide(cam, pts, c2w, depth_np, H, W, fx, fy, cx, cy)
    return np.ascontiguousarray(mask.reshape(X, Y, Z).transpose(2, 1, 0))


def frustum_boundary_points_np(c2w, val_shape, depth_np, bound, H, W, fx, fy, cx, cy):
    """Test support for the frustum mask (tests/test_gpu_mapping.py): WHICH grid points may legitimately come out differently
    in two correct float32 implementations of src/Mapper.py:111-153, as an explicit set.

    The one place where the reference leaves the evaluation order open is the float32 transform `w2c @ homo` (:116-117: numpy's
    batched matmul; the kernel sums left to right): two orders of a four-term float32 sum differ by at most
    6 x 2^-24 x sum |terms| per camera coordinate (each order is within 3 roundings of the exact sum).  Everything after it is
    float64 or a single rounding.  A grid point is a BOUNDARY point when its decision changes somewhere in that box -- i.e. its
    pixel coordinates sit on the image border, its 1/32-pixel map rounding sits on a tie that changes the looked-up depth across
    the depth test, its camera depth sits on 0 or on depth + 0.5 -- or when its squared distance to the camera centre is within
    float32 summation noise of 0.25.  Returns (boundary [Z, Y, X] bool, mask [Z, Y, X] bool)."""
    import numpy as np
    Z, Y, X = val_shape
    pts = _frustum_points(val_shape, bound)
    c2w = c2w.cpu().numpy()
    w2c = np.linalg.inv(c2w)
    P = pts.numpy().astype(np.float32)
    homo = np.concatenate([P, np.ones((P.shape[0], 1), np.float32)], 1)
    cam = (w2c @ homo.reshape(-1, 4, 1))[:, :3]
    mag = (np.abs(w2c[:3, :].astype(np.float64))[None] * np.abs(homo.astype(np.float64))[:, None, :]).sum(-1)   # [n, 3] sum of |terms|
    tol = (6.0 * 2.0 ** -24 * mag).astype(np.float32).reshape(-1, 3, 1)
    dist = pts - torch.from_numpy(c2w[:3, 3]).unsqueeze(0)
    d2 = torch.sum(dist * dist, 1).numpy()
    base, top = _frustum_decide(cam, pts, c2w, depth_np, H, W, fx, fy, cx, cy)
    differs = np.zeros_like(base)
    for sx in (-1.0, 1.0):
        for sy in (-1.0, 1.0):
            for sz in (-1.0, 1.0):
                sgn = np.array([sx, sy, sz], np.float32).reshape(1, 3, 1)
                for s2 in (-1.0, 1.0):
                    m, _ = _frustum_decide(cam + sgn * tol, pts, c2w, depth_np, H, W, fx, fy, cx, cy, dmax=top,
                                           dist2=d2 * (1.0 + s2 * 6.0 * 2.0 ** -24))
                    differs |= m != base
    to_grid = lambda a: np.ascontiguousarray(a.reshape(X, Y, Z).transpose(2, 1, 0))
    return to_grid(differs), to_grid(base)
