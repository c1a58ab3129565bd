"""
Ray sharding over the GPUs of one node (new functionality: the reference has no distributed
code at all).  One process per GPU, torch.distributed with backend "nccl"
(= RCCL over xGMI on ROCm); the same code runs under "gloo" for the CPU tests.

Rays are independent units of the path, so the data path needs NO collective.  The only
batch-global quantity is max(gt_depth) in the far clamp and in the zero-depth surface range
(reference src/utils/Renderer.py:159, :195): it is all-reduced (MAX, one float) BEFORE sharding and
handed to the kernels (adfp_render_args.depth_max) so that a sharded render reproduces the
single-GPU result bit for bit.  Outputs are all-gathered (20 B/ray).  Training adds one
flat-bucket all-reduce (SUM) of the loss gradients per iteration; because the Mapper losses are
plain sums (src/Mapper.py:457-469) the summed shard gradients equal the single-GPU gradient
(up to fp32 summation order) with no rescaling.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous slice [lo, hi) of n units owned by `rank` (sizes differ by at most one)."""
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


def global_depth_max(gt_depth, group=None):
    """max(gt_depth) over every rank's rays, as a 1-element float32 tensor on gt_depth's device."""
    m = gt_depth.detach().reshape(-1).float().max().reshape(1) if gt_depth.numel() else \
        torch.full((1,), float('-inf'), device=gt_depth.device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
    return m


def _all_gather_rows(x, sizes, group):
    """all-gather tensors whose first dim differs per rank (sizes[r] rows on rank r)."""
    return _all_gather_packed((x,), sizes, group)[0]


def _all_gather_packed(outs, sizes, group):
    """ONE collective for several per-ray outputs: every rank packs its rows (depth f64, uncertainty f64, colour 3 x f32 ... =
    28 B per ray for a render) into one row buffer padded to the largest shard, all_gather_into_tensor, unpack.  A frame's
    outputs are ~1 MB per rank at 8 GPUs: the collective is latency bound, so one launch instead of one per output.  On the GPU
    the two sides are ONE kernel each (adfp_gather_pack straight into the send buffer, adfp_gather_unpack out of the gathered
    one; no zero fill, no per-output slice copies, no torch.cat); host tensors (the gloo tests) take the torch composition."""
    world, pad = len(sizes), max(sizes)
    rows = [o.contiguous().reshape(o.shape[0], -1) for o in outs]
    widths = [r.shape[1] * r.element_size() for r in rows]
    dev = outs[0].device
    if dev.type == 'cuda' and all(wd % 4 == 0 for wd in widths) and len(outs) <= 8 and world <= 64:
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        W = sum(widths)
        n = len(outs)
        words = (C.c_int * n)(*[wd // 4 for wd in widths])
        with _lib.device_guard(dev):
            st = _lib.current_stream(dev)
            send = torch.empty((pad, W), dtype=torch.uint8, device=dev)        # rows beyond the shard: never read by unpack
            gathered = torch.empty((world * pad, W), dtype=torch.uint8, device=dev)
            src = (C.c_void_p * n)(*[r.data_ptr() for r in rows])
            _lib.check(L.adfp_gather_pack(n, src, words, rows[0].shape[0], _lib.ptr(send), st), 'adfp_gather_pack')
            dist.all_gather_into_tensor(gathered, send, group=group)
            total = sum(sizes)
            res = [torch.empty((total,) + tuple(o.shape[1:]), dtype=o.dtype, device=dev) for o in outs]
            dst = (C.c_void_p * n)(*[r.data_ptr() for r in res])
            per = (C.c_longlong * world)(*sizes)
            _lib.check(L.adfp_gather_unpack(n, dst, words, world, pad, per, _lib.ptr(gathered), st), 'adfp_gather_unpack')
        return tuple(res)
    buf = torch.zeros((pad, sum(widths)), dtype=torch.uint8, device=dev)
    off = 0
    for r, wd in zip(rows, widths):
        if r.shape[0]:
            buf[:r.shape[0], off:off + wd] = r.view(torch.uint8)
        off += wd
    gathered = torch.empty((world * pad, sum(widths)), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(gathered, buf, group=group)
    gathered = gathered.reshape(world, pad, -1)
    res, off = [], 0
    for o, wd in zip(outs, widths):
        parts = [gathered[r, :sizes[r], off:off + wd] for r in range(world)]
        flat = torch.cat(parts, dim=0).contiguous().view(o.dtype)
        res.append(flat.reshape((flat.shape[0],) + tuple(o.shape[1:])))
        off += wd
    return tuple(res)


def render_rays_sharded(render_fn, rays_o, rays_d, gt_depth, group=None, gather=True):
    """Every rank holds the full ray batch; rank r renders rays shard_range(N, r, world) with the
    full-batch depth max and (optionally) all-gathers the outputs.

    render_fn(rays_o, rays_d, gt_depth, depth_max) -> tuple of tensors with leading dim = #rays
    (e.g. lambda o, d, z, m: renderer.render_batch_ray(c, dec, d, o, dev, tsdf, bnds, stage, z, depth_max=m)).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = rays_o.shape[0]
    lo, hi = shard_range(n, rank, world)
    dmax = None
    if gt_depth is not None:
        # every rank already holds the full depth vector, so the max needs no communication;
        # global_depth_max() is for callers that hold only their own shard
        dmax = gt_depth.detach().reshape(-1).float().max().reshape(1)
    outs = render_fn(rays_o[lo:hi], rays_d[lo:hi], None if gt_depth is None else gt_depth.reshape(-1)[lo:hi], dmax)
    if not gather or world == 1:
        return outs
    sizes = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
    return _all_gather_packed(tuple(outs), sizes, group)


def render_img_sharded(renderer, c, decoders, c2w, device, tsdf_volume, tsdf_bnds, stage, gt_depth, group=None, gather=True):
    """Renderer.render_img with the frame's rays sharded over the ranks: rank r renders the contiguous pixel
    range shard_range(H W, r, world) with the far clamp of the reference's ray batches taken over the whole frame
    (Renderer.render_img_shard), and ONE packed all-gather (28 B per ray) hands every rank the frame -- the images render_img
    returns, bit for bit.  gather=False: the rank's flat shard only."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    H, W = renderer.H, renderer.W
    n = H * W
    lo, hi = shard_range(n, rank, world)
    outs = renderer.render_img_shard(c, decoders, c2w, device, tsdf_volume, tsdf_bnds, stage, gt_depth, lo, hi)
    if not gather:
        return outs
    if world > 1:
        sizes = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
        outs = _all_gather_packed(tuple(outs), sizes, group)
    depth, unc, color = outs
    return depth.reshape(H, W), unc.reshape(H, W), color.reshape(H, W, 3)


def _common_bucket(grads):
    """If every gradient is a contiguous float32 view of ONE storage and together they cover (almost all of) one span of it, that
    span as a flat tensor; otherwise None.  Elements of the span that belong to no listed gradient (frozen parameters of a network
    whose flat gradient was produced anyway) are summed too, harmlessly."""
    if not grads or any(g is None or g.dtype != torch.float32 or not g.is_contiguous() for g in grads):
        return None
    st = grads[0].untyped_storage()
    base = st.data_ptr()
    if any(g.untyped_storage().data_ptr() != base for g in grads):
        return None
    lo = min(g.storage_offset() for g in grads)
    hi = max(g.storage_offset() + g.numel() for g in grads)
    if hi - lo > 1.25 * sum(g.numel() for g in grads) + 64:
        return None
    return grads[0].new_empty(0).set_(st, lo, (hi - lo,))


def allreduce_grads(tensors, group=None, skip_single=True):
    """One flat-bucket all-reduce (SUM, fp32) of the gradients of `tensors` (parameters or grids).
    Tensors without a gradient contribute zeros so that every rank issues the same collective.
    Returns the number of bytes that went through the collective (0 when it was skipped).
    skip_single=False issues the collective even in a world of one rank (bench / RCCL smoke test)."""
    if not (dist.is_available() and dist.is_initialized()) or (skip_single and dist.get_world_size(group) == 1):
        return 0
    tensors = [t for t in tensors if t.requires_grad]
    if not tensors:
        return 0
    span = _common_bucket([t.grad for t in tensors])
    if dist.get_world_size(group) > 1:
        # WHICH path a rank takes depends on local state (did autograd keep the handed views or clone them? did a rank accumulate two
        # backward calls?), and ranks on different paths would issue collectives of different sizes.  So the choice is made
        # collectively: one tiny all-reduce (MIN of [n, -n], n = the span's element count or 0) tells every rank whether ALL ranks
        # hold the same contiguous bucket; otherwise all of them pack.
        n = span.numel() if span is not None else 0
        probe = torch.tensor([n, -n], dtype=torch.int64, device=tensors[0].device)
        dist.all_reduce(probe, op=dist.ReduceOp.MIN, group=group)
        lo, hi = int(probe[0]), -int(probe[1])
        if not (lo == hi == n and n > 0):
            span = None
    if span is not None:
        # The backward of Renderer.render_batch_ray hands autograd views of ONE buffer (engine.render_backward), and autograd
        # keeps them as the .grad tensors: the bucket is contiguous as it stands -- all-reduce it in place, no packing copies
        # (48.6 MB each way for room0's dense grids).
        dist.all_reduce(span, op=dist.ReduceOp.SUM, group=group)
        return span.numel() * 4
    flats = []
    for t in tensors:
        g = t.grad if t.grad is not None else torch.zeros_like(t)
        flats.append(g.reshape(-1).float())
    bucket = torch.cat(flats)
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for t in tensors:
        n = t.numel()
        g = bucket[off:off + n].reshape(t.shape).to(t.dtype)
        if t.grad is None:
            t.grad = g.clone()
        else:
            t.grad.copy_(g)
        off += n
    return bucket.numel() * 4


class MaskedGradBucket:
    """All-reduce of the grid gradients restricted to the frustum-selected voxels (SURVEY.md section 8e,
    "reduce only the frustum-masked subset"): outside the mask the Mapper never updates a grid
    (src/Mapper.py:345-361), so those gradients need not travel.  The voxel index lists are built ONCE per
    mapping call from the masks (identical on every rank, so the compact buckets line up); per iteration the
    masked columns of every grid gradient and the other tensors' gradients go through one flat all-reduce.

        bucket = MaskedGradBucket(c, masks, extra=list(decoders.parameters()))
        loss.backward(); bucket.allreduce(); opt_grids.step(lrs); optimizer.step()
    """

    def __init__(self, grids, masks, extra=(), group=None):
        self.group = group
        self.grids = [(k, g) for k, g in grids.items() if g.requires_grad]
        self.extra = [t for t in extra if t.requires_grad]
        self.index = {}
        for k, g in self.grids:
            m = masks.get(k) if masks is not None else None
            if m is None:
                self.index[k] = None
            else:
                if tuple(m.shape) != tuple(g.shape[2:]):
                    raise ValueError(f'{k}: mask shape {tuple(m.shape)} != grid {tuple(g.shape[2:])}')
                self.index[k] = torch.nonzero(m.reshape(-1).to(g.device), as_tuple=False).reshape(-1)

    def numel(self):
        n = sum(t.numel() for t in self.extra)
        for k, g in self.grids:
            idx = self.index[k]
            n += g.numel() if idx is None else g.shape[1] * idx.numel()
        return n

    @torch.no_grad()
    def allreduce(self, skip_single=True):
        if not (dist.is_available() and dist.is_initialized()) or (skip_single and dist.get_world_size(self.group) == 1):
            return
        parts = []
        for k, g in self.grids:
            gr = g.grad if g.grad is not None else torch.zeros_like(g)
            idx = self.index[k]
            flat = gr.reshape(g.shape[1], -1)
            parts.append((flat if idx is None else flat.index_select(1, idx)).reshape(-1).float())
        for t in self.extra:
            parts.append((t.grad if t.grad is not None else torch.zeros_like(t)).reshape(-1).float())
        if not parts:
            return
        bucket = torch.cat(parts)
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group)
        off = 0
        for k, g in self.grids:
            idx = self.index[k]
            C = g.shape[1]
            n = g.numel() if idx is None else C * idx.numel()
            piece = bucket[off:off + n].reshape(C, -1).to(g.dtype)
            if g.grad is None:
                g.grad = torch.zeros_like(g)
            if idx is None:
                g.grad.copy_(piece.reshape(g.shape))
            else:
                g.grad.reshape(C, -1).index_copy_(1, idx, piece)     # outside the mask: the local gradient, never used
            off += n
        for t in self.extra:
            n = t.numel()
            piece = bucket[off:off + n].reshape(t.shape).to(t.dtype)
            if t.grad is None:
                t.grad = piece.clone()
            else:
                t.grad.copy_(piece)
            off += n
