"""
The Mapper's per-call bookkeeping around the render path, on the MI355X:

``frustum_mask``     = ``Mapper.get_mask_from_c2w`` (reference src/Mapper.py:90-158), which the reference
                       runs on the host with numpy + ``cv2.remap`` once per grid per mapping call;
``MaskedGridAdam``   = the optimisation of the masked part of each feature grid.  The reference keeps a
                       compact copy ``val_grad = val[mask]`` as the Adam parameter and index_puts it into the
                       dense grid before AND after every iteration (src/Mapper.py:347-361, :382-388,
                       :476-482); here the dense grid itself is updated in place on the masked voxels by one
                       kernel per grid -- same values, no boolean-index gathers / scatters, no host sync.

How the reference's ``optimize_map`` would use it (the grids stay ordinary leaf tensors)::

    opt_grids = MaskedGridAdam(c, {k: frustum_mask(cur_c2w, c[k].shape[2:], cur_gt_depth, self.bound,
                                                   H, W, fx, fy, cx, cy) for k in c})
    for joint_iter in range(num_joint_iters):
        ...                                   # render_batch_ray(c, ...) -> loss -> loss.backward()
        opt_grids.step({'grid_low': low_lr, 'grid_high': high_lr, 'grid_color': color_lr})
        optimizer.step()                      # torch Adam over the decoder / mlp parameters only

There is no CPU fallback: everything raises on host tensors.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, ptr, check


def frustum_mask(c2w, val_shape, depth, bound, H, W, fx, fy, cx, cy):
    """Boolean mask ``[Z, Y, X]`` of the grid points selected for optimisation (src/Mapper.py:90-158; the
    reference returns it as ``[X, Y, Z]`` and permutes at the call site, :345).

    c2w [4,4] camera pose, ``val_shape`` = ``val.shape[2:]`` = (Z, Y, X), ``depth`` [H,W] float image on
    the GPU, ``bound`` [3,2]."""
    _lib.require_cuda(depth, 'depth')
    dev = depth.device
    Z, Y, X = (int(v) for v in val_shape)
    m = torch.as_tensor(c2w).detach().to('cpu', torch.float32)
    w2c = torch.linalg.inv(m)                                       # np.linalg.inv(c2w), src/Mapper.py:113
    a_c2w = (C.c_float * 16)(*m.reshape(-1).tolist())
    a_w2c = (C.c_float * 16)(*w2c.reshape(-1).tolist())
    b = _lib.Bound()
    bb = torch.as_tensor(bound).detach().to('cpu', torch.float64)
    for k in range(3):
        b[k][0], b[k][1] = float(bb[k, 0]), float(bb[k, 1])
    with torch.cuda.device(dev):
        d = depth.detach().to(torch.float32).contiguous()
        sampled = torch.empty((X * Y * Z,), dtype=torch.float32, device=dev)
        scratch = torch.empty((1,), dtype=torch.int32, device=dev)
        mask = torch.empty((Z, Y, X), dtype=torch.uint8, device=dev)
        check(lib().adfp_frustum_mask(X, Y, Z, C.byref(b), C.byref(a_c2w), C.byref(a_w2c), float(fx), float(fy), float(cx),
                                      float(cy), int(H), int(W), ptr(d), ptr(sampled), ptr(scratch), ptr(mask),
                                      _lib.current_stream(dev)), 'adfp_frustum_mask')
    return mask.bool()


class MaskedGridAdam:
    """Adam over the masked voxels of the feature grids, in place (see the module docstring).

    ``grids``  dict name -> leaf tensor ``[1, C, Z, Y, X]`` float32 on the GPU (``.grad`` is read by ``step``)
    ``masks``  dict name -> bool/uint8 ``[Z, Y, X]`` (or None = the whole grid, ``frustum_feature_selection: False``)
    State (exp_avg, exp_avg_sq, step count) starts at zero like the fresh optimizer the reference builds in
    every ``optimize_map`` call (src/Mapper.py:374)."""

    def __init__(self, grids, masks=None, betas=(0.9, 0.999), eps=1e-8):
        self.grids = dict(grids)
        self.betas, self.eps = betas, eps
        self.masks, self.state = {}, {}
        for k, g in self.grids.items():
            _lib.require_cuda(g, k)
            if g.dtype != torch.float32 or not g.is_contiguous():
                raise ValueError(f'{k}: expected a contiguous float32 grid')
            mk = None if masks is None else masks.get(k)
            if mk is not None:
                if tuple(mk.shape) != tuple(g.shape[2:]):
                    raise ValueError(f'{k}: mask shape {tuple(mk.shape)} != grid {tuple(g.shape[2:])}')
                mk = mk.to(g.device, torch.uint8).contiguous()
            self.masks[k] = mk
            self.state[k] = [torch.zeros_like(g), torch.zeros_like(g), 0]

    def zero_grad(self):
        for g in self.grids.values():
            g.grad = None

    @torch.no_grad()
    def step(self, lrs):
        """One Adam step per grid with the given learning rates (a grid whose lr is 0 still advances its
        moments, like a torch param group with lr 0); grids without a gradient are skipped like torch does."""
        L = lib()
        for k, g in self.grids.items():
            if g.grad is None:
                continue
            grad = g.grad.contiguous()
            st = self.state[k]
            st[2] += 1
            nvox = g.shape[2] * g.shape[3] * g.shape[4]
            with torch.cuda.device(g.device):
                check(L.adfp_masked_adam(ptr(g), ptr(grad), ptr(st[0]), ptr(st[1]),
                                         ptr(self.masks[k]) if self.masks[k] is not None else None, nvox, g.shape[1],
                                         float(lrs[k]), self.betas[0], self.betas[1], self.eps, st[2],
                                         _lib.current_stream(g.device)), 'adfp_masked_adam')
            torch.autograd.graph.increment_version(g)     # updated through a raw pointer: invalidate the layout caches


# ---------------------------------------------------------------------------------------------------------------------
# One optimisation iteration of Mapper.optimize_map (reference src/Mapper.py:380-482) as ONE call: pre-filter (as a keep
# mask), render forward, the Mapper loss and its cotangents, render backward, Adam on the frustum-masked grid voxels and on
# the trainable decoder parameters -- a fixed kernel sequence with no host read-back, replayed from a HIP graph.
#
# What the reference does per iteration on the host, and what replaces it:
#   boolean-mask compaction of the ray batch (a sync, a data-dependent batch size)  -> adfp_prefilter_mask: keep flags +
#       the max sensor depth of the kept rays; dropped rays are rendered too and masked out of loss and gradients
#   loss assembled from ~10 torch ops with another boolean index                    -> adfp_mapper_loss (one kernel)
#   torch.autograd graph walk                                                        -> a direct adfp_render_backward call
#   torch.optim.Adam over 5 parameter groups (~10 multi-tensor launches)             -> adfp_adam_prep + adfp_masked_adam_dev on
#       the dense grids (in place, masked) and on ONE flat buffer per trainable network (the nn.Parameters become views
#       of it, so state_dict / deepcopy / .parameters() are unchanged)
# ---------------------------------------------------------------------------------------------------------------------
def flatten_parameters(module):
    """Re-home the parameters of `module` in ONE contiguous float32 buffer (state_dict order); every nn.Parameter keeps its
    identity and becomes a view of the buffer.  Returns the buffer."""
    params = list(module.parameters())
    from .decoder import _flat_params
    flat = _flat_params(tuple(params))
    if flat.untyped_storage().data_ptr() == params[0].untyped_storage().data_ptr():
        return flat                              # already one buffer (MLP._apply put them there at .to(device))
    flat = torch.cat([p.detach().reshape(-1).float() for p in params])
    off = 0
    for p in params:
        n = p.numel()
        p.data = flat[off:off + n].view(p.shape)
        off += n
    return flat


class MapperIteration(object):
    """
    it = MapperIteration(renderer, decoders, c, masks, tsdf_volume, tsdf_bnds, stage_lr)
    for joint_iter in range(num_joint_iters):
        ... rays_o, rays_d, gt_depth, gt_color = cat(get_samples(...))        # NOT pre-filtered: the iteration does it
        loss = it.step(rays_o, rays_d, gt_depth, gt_color, stage, warmup)      # device double, no sync

    c           dict of the three feature grids, float32 leaf tensors on the GPU, updated IN PLACE
    masks       dict name -> bool [Z,Y,X] (frustum_mask) or None
    stage_lr    {'low': {'low': lr, 'high': lr, 'color': lr, 'decoders': lr, 'mlp': lr}, 'high': {...}, 'color': {...}}
                (configs/df_prior.yaml:65-83 times lr_factor)
    train       which networks Adam updates: ('color', 'att') = fix_high: True, fix_color: False (src/Mapper.py:364-371)
    """

    def __init__(self, renderer, decoders, c, masks, tsdf_volume, tsdf_bnds, stage_lr, w_color_loss=0.2,
                 train=('color', 'att'), betas=(0.9, 0.999), eps=1e-8, use_graph=True, group=None, distributed=None):
        self.rend, self.dec, self.c = renderer, decoders, c
        self.tsdf, self.tsdf_bnds = tsdf_volume, tsdf_bnds
        self.stage_lr, self.w_color, self.betas, self.eps = stage_lr, float(w_color_loss), betas, eps
        self.use_graph = use_graph
        self.dev = next(iter(c.values())).device
        dev = self.dev
        for k, g in c.items():
            _lib.require_cuda(g, k)
            if g.dtype != torch.float32 or not g.is_contiguous():
                raise ValueError(f'{k}: expected a contiguous float32 grid')
        self.masks = {k: (None if masks is None or masks.get(k) is None else masks[k].to(dev, torch.uint8).contiguous()) for k in c}
        # The grids' optimiser state lives in the kernels' CHANNELS-LAST layout [Z,Y,X,32]: a shadow copy of each grid (what the
        # render kernels read), both Adam moments and the gradient.  adfp_adam_grids_cl steps the shadow, writes the new values
        # through to the reference-layout grid in `c` (what the rest of the system sees) and zeroes the gradient it consumed --
        # per grid and iteration that replaces the forward's re-layout, the backward's re-layout and a zero fill.
        def cl_like(g):
            return torch.zeros((g.shape[2], g.shape[3], g.shape[4], g.shape[1]), dtype=torch.float32, device=dev)
        self.shadow = {k: cl_like(g) for k, g in c.items()}
        self._shadow_version = {k: None for k in c}              # c[k]._version the shadow was last made coherent with
        self.gstate = {k: (cl_like(g), cl_like(g)) for k, g in c.items()}
        self.nets = tuple(train)
        attr = {'low': 'low_decoder', 'high': 'high_decoder', 'color': 'color_decoder', 'att': 'mlp'}
        self.flat = {n: flatten_parameters(getattr(decoders, attr[n])) for n in self.nets}
        decoders._plists = {}                                        # the cached parameter tuples are still the same objects
        renderer._engine.side_lane(self.dev)                         # the backward's second lane exists before any graph is captured
        if getattr(decoders, '_foreign', False):                     # a pickled copy received by the (spawned) Mapper process: this
            decoders.mark_owner()                                    # iteration is the writer of its parameters, so their versions are valid here
        self.fstate = {n: (torch.zeros_like(f), torch.zeros_like(f)) for n, f in self.flat.items()}
        # torch.optim.Adam keeps one step counter PER PARAMETER and advances it only when the parameter has a gradient (the high
        # and colour grids join in later stages): one device counter + derived scalars per group
        self.groups = list(c.keys()) + list(self.nets)
        self.step_count = torch.zeros(len(self.groups), dtype=torch.int32, device=dev)
        self.derived = torch.empty((len(self.groups), 2), dtype=torch.float32, device=dev)
        self._pool = None
        self.loss = torch.zeros(1, dtype=torch.float64, device=dev)
        self._loss_scratch = {}                                          # ray count -> scratch of adfp_mapper_loss_step (its ticket word starts at zero)
        self.bound_dev = torch.as_tensor(renderer.bound).to(dev, torch.float64).contiguous()
        # The Mapper's rays are random pixels of several keyframes (src/Mapper.py:421-436): no two neighbours share a cache line of the
        # TSDF volume.  The iteration reads the corner-block copy (one aligned 32-byte piece per lookup, Engine.tsdf_blocks) when the
        # renderer allows it and it fits; the copy is held HERE (captured graphs carry its address) and re-laid in place when
        # somebody writes the volume (_sync_tsdf_blocks).
        self._cb = renderer._engine.tsdf_blocks(tsdf_volume) if (getattr(renderer, 'tsdf_blocks', False) and tsdf_volume is not None) else None
        self._cb_version = tsdf_volume._version if self._cb is not None else None
        # Ray-sharded iteration (new; the reference has no distributed code): every rank passes ITS rays to step(); the loss
        # gradients are combined by ONE RCCL all-reduce (SUM) of a contiguous bucket -- the backward writes the grid and
        # parameter gradients straight into slices of it, in the order low grid | attention net | high grid | colour net |
        # colour grid (a trained low / high net sits next to its grid), so that a stage reduces exactly the prefix it produced -- and
        # the kept rays' max depth by a
        # one-float all-reduce (MAX).  With frustum masks only the selected voxels' columns travel (_allreduce_gradients).
        # The Mapper losses are plain sums, so no rescaling.
        import torch.distributed as tdist
        self.group = group
        self.distributed = (tdist.is_available() and tdist.is_initialized()) if distributed is None else distributed
        # every gradient a stage can produce has a slot, so that the dense all-reduce of the stage's prefix covers them all
        # (stage low: low grid [+ low net]; stage high: + attention / high nets, high grid; stage color: everything)
        order = [('grid', 'low', 'grid_low'), ('flat', 'low', None), ('flat', 'att', None), ('flat', 'high', None),
                 ('grid', 'high', 'grid_high'), ('flat', 'color', None), ('grid', 'color', 'grid_color')]
        self._bucket_layout, off = [], 0
        for kind, name, key in order:
            if kind == 'flat' and name not in self.nets:
                continue
            n = c[key].numel() if kind == 'grid' else self.flat[name].numel()
            self._bucket_layout.append((kind, name, key, off, n))
            off += n
        self.bucket = torch.zeros(off, dtype=torch.float32, device=dev)      # grid slots: channels-last, kept all-zero between iterations
        self._index = {}
        if self.distributed:
            self._build_index()
        self._graphs = {}
        self._static = {}
        self._versioned = list(c.values()) + [p for n in self.nets for p in getattr(decoders, attr[n]).parameters()]

    def new_frame(self, masks=None):
        """What the reference does at the top of every optimize_map call: a FRESH Adam (src/Mapper.py:374) and a new frustum
        mask per grid (:330-361).  The moments and step counters are zeroed and the masks overwritten in place, so the
        captured graphs stay valid."""
        for m, v in list(self.gstate.values()) + list(self.fstate.values()):
            m.zero_()
            v.zero_()
        self.step_count.zero_()
        if masks is not None:
            for k in self.c:
                if (self.masks[k] is None) != (masks.get(k) is None):
                    raise ValueError(f'{k}: a grid cannot switch between masked and unmasked across frames')
                if self.masks[k] is not None:
                    self.masks[k].copy_(masks[k].to(self.dev, torch.uint8))
            if self.distributed:
                self._build_index()

    def _build_index(self):
        """Distributed mode with frustum masks: the selected voxels' flat indices, once per frame (one host sync, like the
        reference's own boolean indexing at src/Mapper.py:345-361); identical on every rank, so the compact buckets line up."""
        self._index = {k: (None if m is None else torch.nonzero(m.reshape(-1), as_tuple=False).reshape(-1)) for k, m in self.masks.items()}

    def _allreduce_gradients(self, grids, flats, end):
        """SUM over the ranks of what this stage's backward produced.  Without masks: the stage's prefix of the contiguous
        bucket, in place.  With masks: outside the mask a grid is never updated, so only the selected voxels' gradient columns
        travel -- gathered with the parameter gradients into one compact buffer, reduced, scattered back."""
        import torch.distributed as tdist
        key = {'low': 'grid_low', 'high': 'grid_high', 'color': 'grid_color'}
        if all(self._index.get(key[n]) is None for n in grids):
            tdist.all_reduce(self.bucket[:end], op=tdist.ReduceOp.SUM, group=self.group)
            return end * 4
        parts = [f for f in flats.values()]
        for n, g in grids.items():                       # channels-last [Z,Y,X,32]: a selected voxel is one 128-byte row
            idx = self._index.get(key[n])
            parts.append(g.reshape(-1) if idx is None else g.reshape(-1, 32).index_select(0, idx).reshape(-1))
        buf = torch.cat(parts)
        tdist.all_reduce(buf, op=tdist.ReduceOp.SUM, group=self.group)
        off = 0
        for f in flats.values():
            f.copy_(buf[off:off + f.numel()])
            off += f.numel()
        for n, g in grids.items():
            idx = self._index.get(key[n])
            if idx is None:
                g.reshape(-1).copy_(buf[off:off + g.numel()])
                off += g.numel()
            else:
                g.reshape(-1, 32).index_copy_(0, idx, buf[off:off + 32 * idx.numel()].reshape(-1, 32))
                off += 32 * idx.numel()
        return buf.numel() * 4

    # ---- the kernel sequence ------------------------------------------------------------------------------------------
    def invalidate_tsdf(self):
        """The TSDF volume was written by something PyTorch's version counter does not see (another process through CUDA IPC, a
        raw-pointer kernel): the next step() re-lays the corner-block copy out of it.  fusion.TSDFVolume.integrate and in-place torch
        ops bump the version and need no call."""
        self._cb_version = None

    def _sync_tsdf_blocks(self):
        if self._cb is not None and self.tsdf._version != self._cb_version:
            self.rend._engine.refresh_tsdf_blocks(self.tsdf, self._cb)
            self._cb_version = self.tsdf._version

    def _sequence(self, ro, rd, gd, gc, stage, warmup, adam=True):
        L = lib()
        dev, eng, dec, rend = self.dev, self.rend._engine, self.dec, self.rend
        st = _lib.current_stream(dev)
        N = ro.shape[0]
        keep = torch.empty((N,), dtype=torch.uint8, device=dev)
        dmax, prefilter = None, None
        if self.distributed:
            # the far clamp sees the whole batch (Renderer.py:159): the pre-filter is a launch of its own, its maximum all-reduced
            import torch.distributed as tdist
            dmax = torch.empty((1,), dtype=torch.float32, device=dev)
            check(L.adfp_prefilter_mask(ptr(ro), ptr(rd), ptr(gd), N, ptr(self.bound_dev), ptr(keep), ptr(dmax), st), 'adfp_prefilter_mask')
            tdist.all_reduce(dmax, op=tdist.ReduceOp.MAX, group=self.group)
        else:
            prefilter = (self.bound_dev, keep)           # a job of the render call's first launch (adfp_render_args.prefilter_bound)
        self._sync_shadows()
        used = {'low': ('low',), 'high': ('low', 'high', 'att'), 'color': ('low', 'high', 'color', 'att')}[stage]
        need_grid = {k: (k in used) for k in ('low', 'high', 'color')}
        need_flat = {n: (n in used and n in self.nets) for n in ('low', 'high', 'color', 'att')}
        depth, unc, color, weight, aux = eng.render_forward(dec, self.c, ro, rd, gd, self.tsdf, self.tsdf_bnds, rend.bound, stage,
                                                            rend.N_samples, rend.N_surface, rend.lindisp, rend.perturb, None, dmax,
                                                            train=True, need_flat=need_flat, tsdf_blocks=self._cb, prefilter=prefilter)
        S = aux['S']
        la = _lib.AdfpLossArgs()
        la.n_rays, la.S, la.stage, la.warmup, la.w_color_loss = N, S, _lib.STAGE[stage], 1 if warmup else 0, self.w_color
        la.depth, la.color, la.weight = depth.data_ptr(), color.data_ptr(), weight.data_ptr()
        la.gt_depth, la.gt_color, la.keep = gd.data_ptr(), gc.data_ptr(), keep.data_ptr()
        g_depth = torch.empty((N,), dtype=torch.float64, device=dev)
        g_color = torch.empty((N, 3), dtype=torch.float32, device=dev)
        g_weight = torch.empty((N, S), dtype=torch.float32, device=dev) if warmup else None
        la.loss, la.g_depth, la.g_color = self.loss.data_ptr(), g_depth.data_ptr(), g_color.data_ptr()
        la.g_weight = g_weight.data_ptr() if warmup else None
        # ONE launch for the loss, its cotangents, the loss word (written, not accumulated: no zero fill) and -- when an optimiser step
        # follows -- the step counters / bias corrections of torch.optim.Adam for this stage's groups (adfp_mapper_loss_step)
        used_groups = self._stage_groups(stage, need_grid, need_flat) if adam else []
        lrs = (C.c_float * len(self.groups))(*([-1.0] * len(self.groups)))
        for gname, lrv in used_groups:
            lrs[self.groups.index(gname)] = float(lrv)
        scratch = self._loss_scratch.get(N)
        if scratch is None:
            scratch = self._loss_scratch[N] = torch.zeros(int(L.adfp_mapper_loss_scratch_bytes(N)) // 8 + 1, dtype=torch.float64, device=dev)
        # the forward call's f16-range flag (adfp_train_state.counter[8]): a repaired forward means this iteration's gradients are
        # zero by construction -- then nobody steps (parameters, moments and step counters stay as they are)
        skip = C.c_void_p(aux['counter_ptr'] + 32)
        b1, b2 = self.betas
        if N > 0:
            check(L.adfp_mapper_loss_step(C.byref(la), ptr(scratch), scratch.numel() * 8, ptr(self.step_count), ptr(self.derived),
                                          len(self.groups) if adam else 0, lrs, b1, b2, skip, st), 'adfp_mapper_loss_step')
        else:                                   # an empty ray shard: nothing to launch for the loss
            self.loss.zero_()
            if adam:
                check(L.adfp_adam_prep(ptr(self.step_count), ptr(self.derived), len(self.groups), lrs, b1, b2, skip, st), 'adfp_adam_prep')
        out_grids, out_flats, end = {}, {}, 0
        for kind, name, key, off, n in self._bucket_layout:
            view = self.bucket[off:off + n]
            if kind == 'grid' and need_grid[name]:
                g = self.c[key]
                out_grids[name] = view.view(g.shape[2], g.shape[3], g.shape[4], g.shape[1])       # channels-last
                end = off + n
            elif kind == 'flat' and need_flat[name]:
                out_flats[name] = view
                end = off + n
        grids, flats, _ = eng.render_backward(dec, self.c, self.tsdf, self.tsdf_bnds, rend.bound, stage, aux, g_depth, None,
                                              g_color if stage == 'color' else None, g_weight, need_grid, need_flat, ray_keep=keep,
                                              out_grids_cl=out_grids, out_flats=out_flats, grids_prezeroed=True, side_lane=eng.lane_for(stage))
        self.bucket_bytes = end * 4
        if self.distributed:
            import torch.distributed as tdist
            self.bucket_bytes = self._allreduce_gradients(grids, flats, end)
            tdist.all_reduce(self.loss, op=tdist.ReduceOp.SUM, group=self.group)
        if not adam:
            # no optimiser step follows (tests / warm-up before a capture): leave the bucket's grid slots zero for the next call.
            # The caller gets copies in the reference layout.
            ret = {n: g.permute(3, 0, 1, 2).unsqueeze(0).contiguous() for n, g in grids.items()}
            for g in grids.values():
                g.zero_()
            return ret, flats
        lr = self.stage_lr[stage]
        # Adam (src/Mapper.py:374-378, :472): a group whose parameters received no gradient in this stage is skipped by torch
        # (grad is None) and is skipped here; a group with lr 0 still advances its moments
        groups, cl_groups = [], []
        for name, key in (('low', 'grid_low'), ('high', 'grid_high'), ('color', 'grid_color')):
            if name in grids:
                g = self.c[key]
                cl_groups.append((key, g, grids[name], self.gstate[key], self.masks[key], g.shape[2] * g.shape[3] * g.shape[4], lr[name]))
        for n in self.nets:
            if n in flats:
                f = self.flat[n]
                groups.append((n, f, flats[n], self.fstate[n], None, f.numel(), 1, lr['decoders' if n in ('high', 'color') else 'mlp']))
        arr = (_lib.AdfpAdamGroup * max(1, len(groups)))()
        for k, (gname, p, g, (m, v), mask, nvox, ch, lrv) in enumerate(groups):
            a = arr[k]
            a.param, a.grad, a.exp_avg, a.exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
            a.mask = mask.data_ptr() if mask is not None else None
            a.nvox, a.channels = int(nvox), int(ch)
            a.derived = self.derived[self.groups.index(gname)].data_ptr()
        carr = (_lib.AdfpAdamClGroup * max(1, len(cl_groups)))()
        for k, (gname, g, grad, (m, v), mask, nvox, lrv) in enumerate(cl_groups):
            a = carr[k]
            a.param_cl, a.param_cm, a.grad_cl = self.shadow[gname].data_ptr(), g.data_ptr(), grad.data_ptr()
            a.exp_avg_cl, a.exp_avg_sq_cl = m.data_ptr(), v.data_ptr()
            a.mask = mask.data_ptr() if mask is not None else None
            a.nvox = int(nvox)
            a.derived = self.derived[self.groups.index(gname)].data_ptr()
        if cl_groups or groups:                                      # grids and networks: one launch
            check(L.adfp_adam_step(len(cl_groups), C.byref(carr), len(groups), C.byref(arr), b1, b2, self.eps, st), 'adfp_adam_step')
        return grids, flats

    def _stage_groups(self, stage, need_grid, need_flat):
        """(group name, lr) of every parameter group that receives a gradient in this stage (= steps, like torch.optim.Adam, which
        skips parameters whose .grad is None; a group with lr 0 still advances its moments)."""
        lr = self.stage_lr[stage]
        out = [(key, lr[name]) for name, key in (('low', 'grid_low'), ('high', 'grid_high'), ('color', 'grid_color')) if need_grid[name]]
        out += [(n, lr['decoders' if n in ('high', 'color') else 'mlp']) for n in self.nets if need_flat[n]]
        return out

    def _sync_shadows(self):
        """Before a forward: every grid's channels-last shadow is what the engine's layout cache holds for it.  A grid somebody
        else wrote since our last step (its version is not the one we left) is re-laid out into the shadow first."""
        eng = self.rend._engine
        for k, g in self.c.items():
            if self._shadow_version[k] != g._version:
                with torch.cuda.device(self.dev):
                    Z, Y, X = g.shape[2:]
                    check(lib().adfp_relayout_grid(ptr(g), ptr(self.shadow[k]), 32, Z, Y, X, _lib.current_stream(self.dev)), 'adfp_relayout_grid')
                self._shadow_version[k] = g._version
            eng.adopt_grid_cl(k, g, self.shadow[k])

    def _drop_volatile_cache_entries(self):
        self.rend._engine._grid_cache.clear()
        for slot in list(self.dec._packed):
            if slot.split('.')[0] in self.nets:
                del self.dec._packed[slot]

    def _bump_versions(self):
        for t in self._versioned:
            torch.autograd.graph.increment_version(t)        # updated through raw pointers: invalidate (data_ptr, _version) caches
        for k, g in self.c.items():                          # ... except the shadows: the Adam kernel kept them coherent
            self._shadow_version[k] = g._version
            self.rend._engine.adopt_grid_cl(k, g, self.shadow[k])

    @torch.no_grad()
    def input_buffers(self, N):
        """The four static tensors (rays_o [N,3], rays_d [N,3], gt_depth [N], gt_color [N,3]) the captured graphs of an N-ray iteration
        read.  A caller that produces its batch straight into them (``common.get_samples_multi(..., out=it.input_buffers(N))``) and
        passes them to ``step`` saves the copy."""
        st = self._static.get(N)
        if st is None:
            dev = self.dev
            st = self._static[N] = (torch.empty((N, 3), device=dev), torch.empty((N, 3), device=dev), torch.empty((N,), device=dev),
                                    torch.empty((N, 3), device=dev))
        return st

    def step(self, rays_o, rays_d, gt_depth, gt_color, stage, warmup=False):
        """One iteration; returns the loss as a device float64 tensor (reading it synchronises -- do so sparingly)."""
        dev = self.dev
        with torch.cuda.device(dev):
            N = rays_o.shape[0]
            self._sync_tsdf_blocks()
            if not self.use_graph or self.distributed:      # collectives stay out of the graph
                self._sequence(rays_o.float().contiguous(), rays_d.float().contiguous(), gt_depth.float().contiguous(),
                               gt_color.float().contiguous(), stage, warmup)
                self._bump_versions()
                return self.loss
            # A replay never passes through Engine.scene(), so the f16-range word is read here (no sync: an event of the previous
            # replay is seen one step late at worst, and that step repaired itself and skipped its Adam).  A network that tripped
            # runs on the exact kernels from now on: the graphs are keyed on the set of latched networks.
            self.dec.absorb_status()
            self._sync_shadows()             # a grid somebody else wrote since our last step is re-laid out here, outside the graph

            def graph_key():
                return (N, stage, bool(warmup), frozenset(self.dec._exact_latch))
            key = graph_key()
            st = self.input_buffers(N)
            srcs = (rays_o, rays_d, gt_depth, gt_color)
            if all(s_ is d_ for d_, s_ in zip(st, srcs)):
                pass                                            # the caller filled the static buffers itself (common.get_samples_multi(out=...))
            elif all(s_.dtype == torch.float32 and s_.device == d_.device and s_.shape == d_.shape for d_, s_ in zip(st, srcs)):
                torch._foreach_copy_(list(st), list(srcs))      # ONE multi-tensor launch (four copies were ~15 us of GPU time per replay)
            else:
                for dst, src in zip(st, srcs):
                    dst.copy_(src)
            g = self._graphs.get(key)
            if g is None:
                # warm the host-side caches (bounds, workspace) without touching any optimiser state, then capture with the
                # layout / packing caches cold so that their kernels are part of the graph
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    self._sequence(*st, stage, warmup, adam=False)
                torch.cuda.current_stream(dev).wait_stream(side)
                torch.cuda.synchronize(dev)
                self.dec.absorb_status()              # the eager warm-up may have tripped the range guard: capture with that network exact
                key = graph_key()
                # What changes from replay to replay -- the grids' channels-last copies and the TRAINED nets' weight images -- must be
                # rebuilt inside the graph, so those cache entries are dropped before the capture; what does not (the FROZEN nets'
                # images, packed eagerly by the warm-up above into ordinary memory) stays cached.  Nothing the capture allocates
                # may outlive it in a long-lived cache: the graphs share one pool and replay in any order, so a block another
                # graph's replay may write must never be what an eager call (Visualizer, Tracker, Mesher) finds in the cache.
                eng = self.rend._engine
                self._drop_volatile_cache_entries()
                g = torch.cuda.CUDAGraph()
                if self._pool is None:
                    self._pool = torch.cuda.graph_pool_handle()      # the graphs replay one at a time: one pool for all
                try:
                    with eng.private_workspaces():    # the graph gets its OWN workspaces (an eager call may grow and free the shared ones)
                        with torch.cuda.graph(g, pool=self._pool):
                            self._sequence(*st, stage, warmup)
                finally:
                    self._drop_volatile_cache_entries()
                self._graphs[key] = g                 # capturing does not execute: fall through to the first replay
            g.replay()
            self._bump_versions()
            return self.loss
