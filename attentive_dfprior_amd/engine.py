"""
Host-side plumbing between the reference-shaped Python objects (Renderer / DF) and the C ABI of
libadfp.so: builds the ``adfp_scene`` descriptor from PyTorch tensors, owns the scratch
workspace, and caches the two per-call conversions the kernels need

  * feature grids  [1,32,Z,Y,X] (src/DF_Prior.py:243-264)  ->  channels-last [Z,Y,X,32]
  * decoder parameters -> packed MFMA images (decoder.DF.packed_weights)

keyed on ``(data_ptr, _version, shape)`` because the Mapper re-materialises the grids every
iteration (src/Mapper.py:382-388) and the Tracker reads them from another process.

PyTorch here is device memory + streams only; no arithmetic of the hot path runs in torch.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, ptr, check


def _host_bound(t):
    """[3,2] tensor (any device) -> nested python floats (one device sync if on GPU)."""
    return [[float(v) for v in row] for row in t.detach().to('cpu', torch.float64).tolist()]


_ACT_FLOATS = {}
_SLAB_LAYOUTS = {}        # (P, stage, mode, ...) -> carve-up of the training slab (Engine.train_state)
_FLAT_FLOATS = {}


def _flat_floats(name):
    hit = _FLAT_FLOATS.get(name)
    if hit is None:
        L = lib()
        hit = _FLAT_FLOATS[name] = int(L.adfp_attention_flat_floats() if name == 'att' else L.adfp_decoder_flat_floats(_lib.DEC_KIND[name]))
    return hit



def _train_act_floats(name):
    hit = _ACT_FLOATS.get(name)
    if hit is None:
        hit = _ACT_FLOATS[name] = int(lib().adfp_train_act_floats(_lib.DEC_KIND[name]))
    return hit


def math_mode():
    """ADFP_MATH=f16x3 (default): forward decoders on f16 MFMA with a 3-product operand split;
    ADFP_MATH=f32: exact f32-input MFMA everywhere."""
    import os
    m = os.environ.get('ADFP_MATH', 'f16x3')
    if m not in ('f16x3', 'f32'):
        raise RuntimeError(f'ADFP_MATH={m}: expected f16x3 or f32')
    return m


class _PrivateWorkspaces(object):
    def __init__(self, engine):
        self.e = engine

    def __enter__(self):
        self.saved = (self.e._ws, self.e._bws, self.e._gcl)
        self.e._ws = self.e._bws = None
        self.e._gcl = {}
        return self

    def __exit__(self, *exc):
        self.e._ws, self.e._bws, self.e._gcl = self.saved
        return False


_STAGE_NETS = {'low': ('low',), 'high': ('low', 'high', 'att'), 'color': ('low', 'high', 'att', 'color')}
_STAGE_GRIDS = {'low': (('low', 'grid_low'),), 'high': (('low', 'grid_low'), ('high', 'grid_high')),
                'color': (('low', 'grid_low'), ('high', 'grid_high'), ('color', 'grid_color'))}


def _align256(n):
    return (n + 255) & ~255


class Engine(object):
    def __init__(self):
        import os
        # ADFP_BWD_* bits handed to the backward entries (adfp_backward_args.options).  ADFP_SCATTER=cache in the environment of
        # the HOST process selects the in-kernel scatter (kernel A/B runs); the library itself reads no environment.
        self.bwd_options = _lib.BWD_SCATTER_IN_KERNEL if os.environ.get('ADFP_SCATTER', '')[:1] == 'c' else 0
        if os.environ.get('ADFP_WGRAD', '')[:1] == 's':          # weight gradients through the staged two-kernel path (A/B runs)
            self.bwd_options |= _lib.BWD_STAGED_WGRAD
        if os.environ.get('ADFP_WGRAD', '')[:1] == 'o':          # ... inside the chain kernel, the ONE-wave-per-SIMD kernel of round 3-4 (A/B runs)
            self.bwd_options |= _lib.BWD_FUSED_ONE_WAVE
        # which part of the split weight images an INFERENCE call keeps current: 'g' (the 16x16x32 kernels of this library);
        # ADFP_IMAGES=hg in the host's environment keeps both (A/B runs against a library built with -DADFP_LC_32X32)
        self.inference_images = os.environ.get('ADFP_IMAGES', 'g')
        # Test / diagnostic switch: the training state also carries dbg_masks_* buffers, into which the EXACT backward kernels
        # export the ReLU decisions they recomputed (adfp_train_state.dbg_masks_*; relu_masks() decodes them)
        self.export_relu_masks = False
        self._ws = None          # forward scratch, grow-only (stream order makes the reuse safe)
        self._bws = None         # backward scratch, same
        self._gcl = {}           # channels-last gradient scratch per grid (render_backward)
        self._grid_cache = {}    # key name -> (key, channels-last tensor)
        self._tsdf_cb = None     # (key, corner-block copy, pinned source storage) of the last TSDF volume asked for (tsdf_blocks)
        self._bound_cache = {}   # id -> (key, host list)
        # The backward's second lane (adfp_backward_args.side_stream): the spatial sort of the sample points runs beside the backward
        # kernels instead of in front of them.  ADFP_SIDE_LANE=0 in the host's environment keeps everything on one stream (A/B runs).
        self.use_side_lane = os.environ.get('ADFP_SIDE_LANE', '1') not in ('0', 'off', 'false', 'no')
        # ... and for which stages: the lane pays when the backward's main path is longer than the sort (nine launches, ~55 us + two
        # cross-stream hops).  Fused iteration at office0, 5 000 x 64, one stream -> lane: stage low 0.204 -> 0.215 ms, high 0.408 ->
        # 0.390, colour 0.695 -> 0.672 (profiles/r06_ab_side_lane.txt).  ADFP_SIDE_LANE=<stage>[+<stage>] overrides.
        lane_env = os.environ.get('ADFP_SIDE_LANE', '')
        self.side_lane_stages = frozenset(lane_env.split('+')) if lane_env and lane_env[0] in 'lhc' else frozenset(('high', 'color'))
        self._side = {}          # device -> (torch.cuda.Stream, [two torch.cuda.Event])
        self._owed_packs = None  # (job table, count, keep-alive) scene(hand_over_packs=True) leaves for the render call's first launch
        self._owed_relayouts = None   # the same for grid conversions (adfp_render_args.relayout_jobs)

    # ---- caches --------------------------------------------------------------------------
    def _alloc_scratch(self, nbytes, device):
        """A scratch allocation that may give the corner-block TSDF copy back: that copy (8 x the volume, tsdf_blocks) is an
        optimisation, the workspaces are not -- on an out-of-memory error the copy is dropped, the allocator's cache released and the
        allocation tried once more (the affected batches read the plain volume from then on if the copy no longer fits)."""
        try:
            return torch.empty(nbytes, dtype=torch.uint8, device=device)
        except torch.cuda.OutOfMemoryError:
            if self._tsdf_cb is None:
                raise
            import logging
            logging.getLogger('attentive_dfprior_amd').warning(
                'out of device memory allocating %d bytes of scratch: dropping the corner-block TSDF copy (%d bytes)', nbytes,
                self._tsdf_cb[1].numel() * 4)
            self._tsdf_cb = None
            torch.cuda.empty_cache()
            return torch.empty(nbytes, dtype=torch.uint8, device=device)

    def workspace(self, n_points, device):
        need = lib().adfp_workspace_bytes(int(n_points))
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = None
            self._ws = self._alloc_scratch(int(need * 1.25) + 1024, device)
        return self._ws

    def bwd_workspace(self, n_points, device):
        """Scratch of the backward entries (adfp_backward_workspace_bytes: ~190 B per point).  Kept across calls like the forward's:
        a training iteration used to allocate it afresh in every backward."""
        need = lib().adfp_backward_workspace_bytes(int(n_points))
        if self._bws is None or self._bws.numel() < need or self._bws.device != device:
            self._bws = None
            self._bws = self._alloc_scratch(int(need * 1.25) + 1024, device)
        return self._bws

    @staticmethod
    def zero_grad_bucket(c, need_grid, need_flat, dev):
        """The gradient outputs of render_backward for a batch of ZERO rays: the same single-allocation layout (grids first, then
        the flat parameter gradients), zero-filled -- a rank whose ray shard is empty then contributes to dist.allreduce_grads
        through the same in-place path as every other rank."""
        sizes = [('g' + n, c[k].numel()) for n, k in (('low', 'grid_low'), ('high', 'grid_high'), ('color', 'grid_color')) if need_grid.get(n)]
        sizes += [('f' + n, _flat_floats(n)) for n in ('low', 'high', 'color', 'att') if need_flat.get(n)]
        bucket = torch.zeros((sum(v for _, v in sizes),), dtype=torch.float32, device=dev)
        grids, flats, off = {}, {}, 0
        for tag, nfl in sizes:
            if tag[0] == 'g':
                key = 'grid_' + tag[1:]
                grids[tag[1:]] = bucket[off:off + nfl].view(c[key].shape)
            else:
                flats[tag[1:]] = bucket[off:off + nfl]
            off += nfl
        return grids, flats

    def _cl_scratch(self, name, shape, device):
        """Channels-last scratch the backward scatters a grid's gradient into before it is re-laid out for autograd (a temporary of
        the call, kept across calls like the workspaces)."""
        t = self._gcl.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.device != device:
            t = self._gcl[name] = torch.empty(shape, dtype=torch.float32, device=device)
        return t

    def lane_for(self, stage):
        """Does a backward of `stage` take the side lane (when its caller asks for one at all)?"""
        return self.use_side_lane and stage in self.side_lane_stages

    def side_lane(self, device):
        """(stream, two events) of the backward's second lane on `device`, created on first use -- never inside a stream capture
        (the events are materialised by a first record, which a capture would swallow): a holder that captures graphs asks for the
        lane BEFORE capturing (mapping.MapperIteration does); None when switched off or when first asked for during a capture."""
        if not self.use_side_lane:
            return None
        key = torch.device(device)
        hit = self._side.get(key)
        if hit is None:
            if torch.cuda.is_current_stream_capturing():
                return None
            with torch.cuda.device(key):
                st = torch.cuda.Stream(device=key)
                evs = [torch.cuda.Event() for _ in range(2)]
                for ev in evs:
                    ev.record(torch.cuda.current_stream(key))           # creates the hipEvent_t behind .cuda_event
            hit = self._side[key] = (st, evs)
        return hit

    def private_workspaces(self):
        """`with engine.private_workspaces():` -- a HIP-graph capture gets workspaces of its OWN (allocated from the graph's pool
        inside the capture), and the eager ones come back afterwards: an eager call may grow and free the shared buffers."""
        return _PrivateWorkspaces(self)

    def host_bound(self, t, slot):
        key = (t.data_ptr(), t._version, t.device)
        hit = self._bound_cache.get(slot)
        if hit is not None and hit[0] == key:
            return hit[1]
        val = _host_bound(t)
        self._bound_cache[slot] = (key, val, t.untyped_storage())     # pins the storage, see grid_cl
        return val

    def adopt_grid_cl(self, name, g, shadow):
        """`shadow` IS the channels-last copy of grid `g` as it stands (the caller keeps it coherent: mapping.MapperIteration updates
        both in one Adam kernel): the next scene() finds it in the cache instead of re-laying the grid out."""
        key = (g.data_ptr(), g._version, g.shape, g.stride())
        self._grid_cache[name] = (key, shadow, g.untyped_storage(), 'adopted')      # never recycled as a re-layout destination

    def grid_cl(self, name, g, defer=None):
        """channels-last copy of a [1,32,Z,Y,X] grid, converted by adfp_relayout_grid -- or, with `defer` (a list), recorded there as
        a job (src, dst, voxels, keep-alive) for ONE launch of the caller's (flush_relayout_jobs / adfp_render_args.relayout_jobs); the
        cache calls the copy current at once, so a caller that fails to run the jobs must drop the entry (scene() does)."""
        _lib.require_cuda(g, name)
        if g.dim() != 5 or g.shape[0] != 1 or g.shape[1] != 32:
            raise RuntimeError(f'{name}: expected [1,32,Z,Y,X], got {tuple(g.shape)}')
        key = (g.data_ptr(), g._version, g.shape, g.stride())
        hit = self._grid_cache.get(name)
        if hit is not None and hit[0] == key:
            return hit[1]
        src = g.detach()
        if src.dtype != torch.float32:
            src = src.float()
        src = src.contiguous()
        Z, Y, X = src.shape[2:]
        if hit is not None and len(hit) == 3 and hit[1].shape == (Z, Y, X, 32) and hit[1].device == src.device:
            dst = hit[1]
        else:
            dst = torch.empty((Z, Y, X, 32), dtype=torch.float32, device=src.device)
        if defer is not None and len(defer) < _lib.RELAYOUT_MAX_JOBS:
            defer.append((src, dst, Z * Y * X))
        else:
            check(lib().adfp_relayout_grid(ptr(src), ptr(dst), 32, Z, Y, X, _lib.current_stream(src.device)),
                  'adfp_relayout_grid')
        # the cache entry keeps the source's storage alive: (data_ptr, _version) identifies the contents only as
        # long as the allocator cannot hand the same address to another tensor
        self._grid_cache[name] = (key, dst, g.untyped_storage())
        return dst

    def tsdf_blocks(self, tsdf_volume):
        """The CORNER-BLOCK copy of a TSDF volume (adfp_relayout_tsdf: [X][Y][Z][8] float32, one aligned 32-byte piece per
        trilinear lookup), built once per volume and cached on (data_ptr, _version, shape, strides) with the source's storage
        pinned -- the TSDF is static for a whole run (get_tsdf.py writes it once, src/DF_Prior.py:86-91 loads it).  8 x the
        volume's bytes (room0: 6.3 GB, the 1024^3 stress volume: 34 GB of the 288 GB); None when that does not fit in half of
        the free memory."""
        t = tsdf_volume
        _lib.require_cuda(t, 'tsdf_volume')
        if t.dtype != torch.float32 or t.dim() != 5 or t.shape[0] != 1 or t.shape[1] != 1:
            return None
        key = (t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()))
        hit = self._tsdf_cb
        if hit is not None and hit[0] == key:
            return hit[1]
        Z, Y, X = t.shape[2:]
        need = 32 * Z * Y * X
        self._tsdf_cb = None                                  # a stale copy goes first
        dev = t.device
        free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        if need > free // 2:
            return None
        import logging
        logging.getLogger('attentive_dfprior_amd').info(
            'corner-block TSDF copy: allocating %.2f GB (8 x the %dx%dx%d volume) on %s; Renderer.tsdf_blocks = False / ADFP_TSDF_BLOCKS=0 '
            'switches it off', need / 1e9, X, Y, Z, dev)
        with _lib.device_guard(dev):
            cb = torch.empty((X, Y, Z, 8), dtype=torch.float32, device=dev)
            td = _lib.AdfpTsdf()
            self.fill_tsdf(td, t, [])
            check(lib().adfp_relayout_tsdf(C.byref(td), ptr(cb), _lib.current_stream(dev)), 'adfp_relayout_tsdf')
        self._tsdf_cb = (key, cb, t.untyped_storage())
        return cb

    def invalidate_tsdf_blocks(self):
        """Drops the cached corner-block copy: for a TSDF volume somebody wrote WITHOUT PyTorch seeing it (a raw-pointer kernel of the
        caller's own, another process through CUDA IPC -- version counters are process-local).  fusion.TSDFVolume.integrate bumps the
        volume's version itself and needs no call.  The next incoherent batch rebuilds the copy (8 x the volume, one kernel)."""
        self._tsdf_cb = None

    def refresh_tsdf_blocks(self, tsdf_volume, cb):
        """Re-lay `tsdf_volume` into an existing corner-block copy `cb` IN PLACE (a holder whose captured graphs carry cb's address:
        MapperIteration after somebody wrote the volume)."""
        with _lib.device_guard(tsdf_volume.device):
            td = _lib.AdfpTsdf()
            self.fill_tsdf(td, tsdf_volume, [])
            check(lib().adfp_relayout_tsdf(C.byref(td), ptr(cb), _lib.current_stream(tsdf_volume.device)), 'adfp_relayout_tsdf')

    @staticmethod
    def relayout_jobs_array(rjobs):
        """(ctypes array, count, keep-alive) of deferred grid conversions (grid_cl(defer=...))."""
        arr = (_lib.AdfpRelayoutJob * len(rjobs))()
        for k, (src, dst, vox) in enumerate(rjobs):
            arr[k].src, arr[k].dst, arr[k].voxels = src.data_ptr(), dst.data_ptr(), vox
        return arr, len(rjobs), [t for j in rjobs for t in j[:2]]

    def flush_relayout_jobs(self, rjobs, device):
        if not rjobs:
            return
        arr, n, _ = self.relayout_jobs_array(rjobs)
        with _lib.device_guard(device):
            check(lib().adfp_relayout_grids(n, arr, 0, _lib.current_stream(device)), 'adfp_relayout_grids')

    # ---- descriptor ----------------------------------------------------------------------
    def scene(self, decoders, c, tsdf_volume, tsdf_bnds, bound, stage, backward=False, ht_nets=(), keys=None, images=None, state=None,
              tsdf_blocks=False, hand_over_packs=False):
        """Returns (AdfpScene, keepalive list).  Forward: every network of the stage takes its f16-split image (ADFP_MATH=f16x3 and
        not latched to exact, DF.uses_split) or its exact f32 image, plus -- when any split image is in use -- the flat parameters
        the device-side f32 repair path needs (adfp_scene.flat_*).  Backward: the exact f32 images, except for the decoders named
        in `ht_nets`: those get their transposed f16-split image (the caller has checked that the forward left their ReLU masks).
        keys: {net: DF.net_key(net)} when the caller has them (a backward re-uses its forward's: parameters must not change in
        between); filled in for the stage's networks otherwise.
        images: which part(s) of the split images to keep current -- None: exactly what the forward entries read for this stage,
        these latches and this training `state` (Engine.image_parts); 'hg' = both (bench / A-B tools that launch single kernels
        of either family)."""
        if state is None:                        # (a training forward absorbed BEFORE it laid its state out: the latch that chose the
            decoders.absorb_status()             # state's regions must be the latch that picks the images below)
        sc = _lib.AdfpScene()                    # an f16-range event of an EARLIER call: that network is exact from now on
        sc.status = decoders.status_word().data_ptr()
        keep = []
        _lib.fill_bound(sc.bound, self.host_bound(bound, 'bound'))
        rjobs = []                               # grids to convert: ONE launch (inside the render call, or flush_relayout_jobs below)
        nets = _STAGE_NETS[stage]
        if keys is None:
            keys = {}
        split = math_mode() == 'f16x3'
        latch = decoders._exact_latch
        any_split = False
        from .decoder import flush_pack_jobs
        jobs = decoders._pack_jobs = []          # the split / transposed images this call has to (re)build: packed in ONE launch below
        try:
            for field, key in _STAGE_GRIDS[stage]:
                g = self.grid_cl(key, c[key], defer=rjobs)
                keep.append(g)
                gd = getattr(sc, field)
                gd.data = g.data_ptr()
                gd.Z, gd.Y, gd.X = g.shape[0], g.shape[1], g.shape[2]
            for n in nets:
                k = keys.get(n)
                if k is None:
                    k = keys[n] = decoders.net_key(n)
                if backward:
                    if n in ht_nets:
                        setattr(sc, 'ht_' + n, decoders.packed_weights(n, 'ht', k).data_ptr())
                    else:
                        setattr(sc, 'w_' + n, decoders.packed_weights(n, 'f32', k).data_ptr())
                elif split and n not in latch:
                    setattr(sc, 'h_' + n, decoders.packed_weights(n, images or self.image_parts(stage, n, latch, state), k).data_ptr())
                    any_split = True
                    # a TRAINING forward whose state has mask room for the network: its backward will read the transposed image --
                    # packed now, in the same launch (the backward finds it current: the parameters do not change in between)
                    if state is not None and ('masks_' + n) in state and not state.get('bwd_exact'):
                        decoders.packed_weights(n, 'ht', k)
                else:
                    setattr(sc, 'w_' + n, decoders.packed_weights(n, 'f32', k).data_ptr())
            # hand_over_packs: the caller's render call packs them in its first launch (adfp_render_args.pack_jobs); at most
            # ADFP_PACK_MAX_JOBS fit one table
            self._owed_relayouts = None
            if rjobs:
                if hand_over_packs:
                    self._owed_relayouts = self.relayout_jobs_array(rjobs)
                else:
                    self.flush_relayout_jobs(rjobs, next(iter(c.values())).device)
                rjobs = []
            self._owed_packs = None
            if hand_over_packs and 0 < len(jobs) <= 8:
                from .decoder import pack_jobs_array
                arr, alive = pack_jobs_array(jobs)
                self._owed_packs = (arr, len(jobs), alive)
                del jobs[:]
            else:
                flush_pack_jobs(jobs, decoders.status_word(), next(iter(c.values())).device)
            if any_split:
                for n in nets:
                    setattr(sc, 'flat_' + n, decoders.flat_weights(n, keys[n]).data_ptr())
            if stage != 'low':
                _lib.fill_bound(sc.tsdf_bnds, self.host_bound(tsdf_bnds, 'tsdf_bnds'))
                self.fill_tsdf(sc.tsdf, tsdf_volume, keep)
                if tsdf_blocks is not None and tsdf_blocks is not False and not backward:
                    # True: this engine's cached copy (None: not float32 / no room -- the plain volume serves); a tensor: the caller's own copy
                    cb = tsdf_blocks if isinstance(tsdf_blocks, torch.Tensor) else self.tsdf_blocks(tsdf_volume)
                    if cb is not None:
                        sc.tsdf.corner_blocks = cb.data_ptr()
                        keep.append(cb)
        except BaseException:
            # packed_weights(defer=...) has already recorded the deferred images as current in decoders._packed; if this call dies
            # before they are packed (still in `jobs`) or handed to a render call (self._owed_packs), the cache would vouch for images
            # nobody wrote: forget all of them, the next call re-packs
            if jobs or self._owed_packs is not None:
                decoders._packed = {}
            if rjobs or self._owed_relayouts is not None:      # the same for channels-last copies nobody wrote
                self._grid_cache = {k: v for k, v in self._grid_cache.items() if len(v) == 4}
            self._owed_packs = self._owed_relayouts = None
            raise
        finally:
            decoders._pack_jobs = None
        return sc, keep

    # ---- training state --------------------------------------------------------------------
    def image_parts(self, stage, net, latch, state):
        """Which part of `net`'s split image the forward entries (adfp_render_forward / adfp_eval_points_train, include/adfp.h at
        adfp_pack_split_image) read, given that `net` itself is f16-split:
          high decoder / attention MLP: a training call whose state has ReLU-mask room for the network -> H (the mask-leaving
                  32x32x16 kernels), otherwise -> G (k_decode_high_g / k_attention_g)
          low / colour decoder: inside the fused low + colour launch -- stage colour, both split, and either no training state
                  (k_decode_lc16) or mask room for BOTH (k_decode_lc16_train) -> G; on their own (stages low / high, the other
                  one latched to exact, a training call with mask room for one of them only) -> H (k_decode_h)
        self.inference_images replaces 'g' (ADFP_IMAGES=hg: a library built with -DADFP_LC_32X32 reads H everywhere)."""
        has = (lambda n: ('masks_' + n) in state) if state is not None else (lambda n: False)
        if net in ('high', 'att'):
            return 'h' if has(net) else self.inference_images
        fused = stage == 'color' and 'low' not in latch and 'color' not in latch and \
            (state is None or (has('low') and has('color')))
        return self.inference_images if fused else 'h'

    @staticmethod
    def train_state(P, stage, dev, decoders, need_flat=None, extra=()):
        """The caller-owned buffers a training forward leaves for the backward (adfp_train_state): always the TSDF stage's
        flags / in-band list / attention inputs; in f16x3 mode also the ReLU masks of every decoder of the stage and, for a
        decoder whose parameter gradient will be asked for (need_flat[name], default: any of its parameters requires
        grad), its layer inputs -- with those the backward runs on f16 MFMA and recomputes nothing.
        ONE allocation (`slab`, uint8) carved at 256-byte boundaries: `offsets[name]` = byte offset, `ptrs[name]` = device address;
        `extra` = further (name, bytes) regions the caller wants in the same slab (render_forward: z_vals, raw)."""
        split = math_mode() == 'f16x3'
        latch = decoders._exact_latch
        if need_flat is None:
            need_flat = {n: any(p.requires_grad for p in decoders.net_params(n)) for n in _STAGE_NETS[stage]}
        lkey = (P, stage, split, tuple(sorted(latch)), tuple(bool(need_flat.get(n)) for n in ('low', 'high', 'color', 'att')), extra)
        lay = _SLAB_LAYOUTS.get(lkey)
        if lay is None:
            regions = [('counter', 64), ('flags', P), ('list', 4 * P), ('att_occ', 4 * P), ('att_u', 4 * P)]
            regions += list(extra)
            nets = ['low'] + (['high'] if stage != 'low' else []) + (['color'] if stage == 'color' else [])
            for n in nets:
                if not split or n in latch:                    # ADFP_MATH=f32, or latched to exact: the exact backward recomputes
                    continue
                regions.append(('masks_' + n, 4 * _lib.TRAIN_MASK_WORDS * P))
                if need_flat.get(n):
                    regions.append(('act_' + n, 4 * _train_act_floats(n) * P))
            if stage != 'low' and split and 'att' not in latch:   # the attention network, rows = in-band list entries (at most P)
                regions.append(('masks_att', 4 * _lib.TRAIN_ATT_MASK_WORDS * P))
                if need_flat.get('att'):
                    regions.append(('act_att', 4 * _lib.TRAIN_ATT_ACT_FLOATS * P))
            offsets, off = {}, 0
            for name, nbytes in regions:
                offsets[name] = off
                off += _align256(nbytes)
            if len(_SLAB_LAYOUTS) > 256:
                _SLAB_LAYOUTS.clear()
            lay = _SLAB_LAYOUTS[lkey] = (offsets, dict(regions), max(off, 256), tuple(n for n, _ in regions if n not in ('z_vals', 'raw')),
                                         {name: True for name in offsets})
        offsets, sizes, total, fields, present = lay
        slab = torch.empty((total,), dtype=torch.uint8, device=dev)
        base = slab.data_ptr()
        ptrs = {name: base + o for name, o in offsets.items()}
        st = _lib.AdfpTrainState()
        for name in fields:
            setattr(st, name, ptrs[name])
        bufs = dict(present)                              # presence flags (Engine.ht_nets asks for 'masks_<net>' / 'act_<net>')
        bufs['slab'], bufs['offsets'], bufs['ptrs'], bufs['sizes'] = slab, offsets, ptrs, sizes
        bufs['counter_ptr'], bufs['bwd_exact'], bufs['_state'] = ptrs['counter'], 'bwd' in latch, st
        bufs['exact_nets'] = frozenset(latch)             # networks whose forward runs on the exact kernels: they leave no masks
        return bufs

    @staticmethod
    def state_tensor(saved, name, dtype, shape=None):
        """A region of the training slab as a tensor (tests / diagnostics: e.g. the ReLU mask words 'masks_color' as int32)."""
        nbytes = saved['sizes'][name]
        t = saved['slab'][saved['offsets'][name]: saved['offsets'][name] + nbytes].view(dtype)
        return t if shape is None else t.view(shape)

    @staticmethod
    def relu_masks(saved, stage):
        """Decodes the ReLU decisions the BACKWARD of a training call differentiated along (diagnostics / parity tests; needs
        Engine.export_relu_masks for the networks that took the exact backward): {'low' / 'high' / 'color': bool [P, 5, 32] (layer,
        unit; for 'high' only the in-band points are meaningful, 'high_valid' bool [P] marks them), 'att': [bool [P, 64], [P, 128],
        [P, 128], [P, 64]] (rows of points outside the band are False), 'band': bool [P]}.  A network on the f16-split backward
        used the masks its training forward left (masks_<net>); one on the exact backward exported what it recomputed
        (dbg_masks_<net>, adfp_train_state)."""
        P = saved['sizes']['flags']
        dev = saved['slab'].device
        ht = saved.get('ht_used', ())
        count = int(Engine.state_tensor(saved, 'counter', torch.int32)[0]) if stage != 'low' else 0
        lst = Engine.state_tensor(saved, 'list', torch.int32)[:count].long() if stage != 'low' else None
        u = torch.arange(32, device=dev)
        half, bit = (u >> 2) & 1, 15 - (4 * (u >> 3) + (u & 3))                   # unit -> lane half, bit of the 16-bit layer field
        out = {}
        for n in _STAGE_NETS[stage]:
            src = ('masks_' if n in ht else 'dbg_masks_') + n
            if src not in saved['sizes']:
                raise RuntimeError(f'relu_masks: no {src} in the training state (set Engine.export_relu_masks before the forward)')
            if n == 'att':
                w = Engine.state_tensor(saved, src, torch.int32, (P, 2, 7))[:count].long() & 0xFFFFFFFF
                layers = []
                for first, units in ((0, 64), (1, 128), (3, 128), (5, 64)):
                    uu = torch.arange(units, device=dev)
                    hh = (uu >> 2) & 1
                    v = 16 * (uu >> 5) + 4 * ((uu & 31) >> 3) + (uu & 3)          # value index inside the layer: 16 ob + r
                    word, b = first + (v >> 5), 31 - (v & 31)
                    m = ((w[:, hh, word] >> b) & 1).bool()                         # [count, units]
                    full = torch.zeros((P, units), dtype=torch.bool, device=dev)
                    full[lst] = m
                    layers.append(full)
                out['att'] = layers
                out['att_softmax'] = w[:, :, 6]
                continue
            w = Engine.state_tensor(saved, src, torch.int32, (P, 2, 3)).long() & 0xFFFFFFFF
            rows = w[:count] if n == 'high' else w
            m = torch.stack([((rows[:, half, i >> 1] >> (16 * (i & 1) + bit)) & 1).bool() for i in range(5)], 1)     # [rows, 5, 32]
            if n == 'high':
                full = torch.zeros((P, 5, 32), dtype=torch.bool, device=dev)
                full[lst] = m
                valid = torch.zeros((P,), dtype=torch.bool, device=dev)
                valid[lst] = True
                out['high'], out['high_valid'] = full, valid
            else:
                out[n] = m
        if stage != 'low':
            band = torch.zeros((P,), dtype=torch.bool, device=dev)
            band[lst] = True
            out['band'] = band
        return out

    @staticmethod
    def ht_nets(saved, need_flat, need_pos, need_grid=None):
        """Decoders whose backward can run f16-split: the forward left their masks, and their layer inputs if their parameter
        gradient is wanted.  A position / ray gradient comes from the f16-split kernels only for a network whose own parameter
        and grid gradients are NOT wanted (the Tracker: everything frozen but the pose); bundle adjustment takes the exact ones."""
        if saved.get('bwd_exact'):
            return ()
        need_grid = need_grid or {}

        exact = saved.get('exact_nets', ())

        def ok(n):
            if n in exact or ('masks_' + n) not in saved or (need_flat.get(n) and ('act_' + n) not in saved):
                return False
            return not (need_pos and (need_flat.get(n) or need_grid.get(n)))
        return tuple(n for n in ('low', 'high', 'color', 'att') if ok(n))

    @staticmethod
    def fill_tsdf(td, tsdf_volume, keep):
        _lib.require_cuda(tsdf_volume, 'tsdf_volume')
        t = tsdf_volume
        if t.dtype != torch.float32:
            t = t.float()
            keep.append(t)
        if t.dim() != 5 or t.shape[0] != 1 or t.shape[1] != 1:
            raise RuntimeError(f'tsdf_volume: expected [1,1,Z,Y,X], got {tuple(t.shape)}')
        td.data = t.data_ptr()
        td.Z, td.Y, td.X = t.shape[2], t.shape[3], t.shape[4]
        td.sZ, td.sY, td.sX = t.stride(2), t.stride(3), t.stride(4)

    # ---- a5..a12 ---------------------------------------------------------------------------
    def eval_points(self, decoders, pts, c, tsdf_volume, tsdf_bnds, bound, stage, apply_bound_rule=True):
        """pts [P,3] f64/f32 on the GPU -> raw [P,4] f32, w [P] f32.  Autograd-transparent like the reference's
        Renderer.eval_points / DF.forward (src/utils/Renderer.py:27-71): with grad enabled and a point tensor, a grid or a
        decoder parameter that requires grad, the call goes through adfp_eval_points_train / adfp_eval_points_backward."""
        _lib.require_cuda(pts, 'points')
        if torch.is_grad_enabled() and (pts.requires_grad or any(v.requires_grad for v in c.values()) or decoders.any_requires_grad()):
            from .autograd import eval_points_with_grad
            return eval_points_with_grad(self, decoders, pts, c, tsdf_volume, tsdf_bnds, bound, stage, apply_bound_rule)
        raw, w, _ = self.eval_points_forward(decoders, pts, c, tsdf_volume, tsdf_bnds, bound, stage, apply_bound_rule)
        return raw, w

    def eval_points_forward(self, decoders, pts, c, tsdf_volume, tsdf_bnds, bound, stage, apply_bound_rule=True, train=False,
                            need_flat=None):
        dev = pts.device
        saved = None
        with _lib.device_guard(dev):
            if pts.dtype == torch.float64:
                mode = _lib.PTS_F64
            else:
                mode = _lib.PTS_F32
                if pts.dtype != torch.float32:
                    pts = pts.float()
            pts = pts.detach().contiguous()
            P = pts.shape[0]
            raw = torch.empty((P, 4), dtype=torch.float32, device=dev)
            w = torch.empty((P,), dtype=torch.float32, device=dev)
            if P == 0:
                return raw, w, saved
            keys = {}
            if train:
                decoders.absorb_status()          # BEFORE the state is laid out: a network that latches to exact now gets no mask room
                saved = self.train_state(P, stage, dev, decoders, need_flat)
            sc, keep = self.scene(decoders, c, tsdf_volume, tsdf_bnds, bound, stage, keys=keys, state=saved)
            ap = _lib.AdfpPoints()
            ap.mode = mode
            ap.n_points = P
            ap.pts = pts.data_ptr()
            ws = self.workspace(P, dev)
            st = None
            if train:
                saved.update(pts=pts, mode=mode, keys=keys)
                st = saved['_state']
            check(lib().adfp_eval_points_train(C.byref(sc), C.byref(ap), _lib.STAGE[stage], 1 if apply_bound_rule else 0,
                                               ptr(raw), ptr(w), ptr(ws), ws.numel(), C.byref(st) if st is not None else None,
                                               _lib.current_stream(dev)), 'adfp_eval_points')
        return raw, w, saved

    def eval_points_backward(self, decoders, c, tsdf_volume, tsdf_bnds, bound, stage, apply_bound_rule, saved, g_raw, g_w,
                             need_grid, need_flat, need_pts):
        """Returns (grid grads dict in the reference's [1,32,Z,Y,X] layout, flat parameter grads dict, d/d pts [P,3] f32 or None)."""
        pts = saved['pts']
        dev = pts.device
        L = lib()
        with _lib.device_guard(dev):
            P = pts.shape[0]
            sc, keep = self.scene(decoders, c, tsdf_volume, tsdf_bnds, bound, stage, backward=True,
                                  ht_nets=self.ht_nets(saved, need_flat, need_pts, need_grid), keys=saved.get('keys'))
            ap = _lib.AdfpPoints()
            ap.mode, ap.n_points, ap.pts = saved['mode'], P, pts.data_ptr()
            a = _lib.AdfpPointsBackwardArgs()
            a.stage, a.flags, a.state = _lib.STAGE[stage], 1 if apply_bound_rule else 0, saved['_state']
            graw = None if g_raw is None else g_raw.detach().to(dev, torch.float32).contiguous()
            gw = None if g_w is None else g_w.detach().to(dev, torch.float32).contiguous()
            a.g_raw, a.g_w = _lib.ptr(graw), _lib.ptr(gw)
            grids_cl, flats = {}, {}
            for name, key in (('low', 'grid_low'), ('high', 'grid_high'), ('color', 'grid_color')):
                if need_grid.get(name):
                    Z, Y, X = c[key].shape[2:]
                    grids_cl[name] = torch.empty((Z, Y, X, 32), dtype=torch.float32, device=dev)
                    setattr(a, 'g_grid_' + name, grids_cl[name].data_ptr())
            for name in ('low', 'high', 'color', 'att'):
                if need_flat.get(name):
                    flats[name] = torch.empty((_flat_floats(name),), dtype=torch.float32, device=dev)
                    setattr(a, 'g_flat_' + name, flats[name].data_ptr())
            g_pts = torch.empty((P, 3), dtype=torch.float32, device=dev) if need_pts else None
            a.g_pts = _lib.ptr(g_pts)
            ws = self.bwd_workspace(P, dev)
            a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
            a.options = self.bwd_options
            stream = _lib.current_stream(dev)
            check(L.adfp_eval_points_backward(C.byref(sc), C.byref(ap), C.byref(a), stream), 'adfp_eval_points_backward')
            grids = {}
            if grids_cl:
                arr = (_lib.AdfpRelayoutJob * len(grids_cl))()
                for k, (name, g) in enumerate(grids_cl.items()):
                    Z, Y, X = g.shape[:3]
                    out = torch.empty((1, 32, Z, Y, X), dtype=torch.float32, device=dev)
                    arr[k].src, arr[k].dst, arr[k].voxels = g.data_ptr(), out.data_ptr(), Z * Y * X
                    grids[name] = out
                check(L.adfp_relayout_grids(len(grids_cl), arr, 1, stream), 'adfp_relayout_grids')
        return grids, flats, g_pts

    def sample_tsdf(self, pts, tsdf_volume, tsdf_bnds):
        _lib.require_cuda(pts, 'points')
        dev = pts.device
        with _lib.device_guard(dev):
            mode = _lib.PTS_F64 if pts.dtype == torch.float64 else _lib.PTS_F32
            if mode == _lib.PTS_F32 and pts.dtype != torch.float32:
                pts = pts.float()
            pts = pts.detach().reshape(-1, 3).contiguous()
            P = pts.shape[0]
            out = torch.empty((P,), dtype=torch.float32, device=dev)
            if P == 0:
                return out
            td = _lib.AdfpTsdf()
            keep = []
            self.fill_tsdf(td, tsdf_volume, keep)
            b = _lib.Bound()
            _lib.fill_bound(b, self.host_bound(tsdf_bnds, 'tsdf_bnds'))
            ap = _lib.AdfpPoints()
            ap.mode = mode
            ap.n_points = P
            ap.pts = pts.data_ptr()
            check(lib().adfp_sample_tsdf(C.byref(td), C.byref(b), C.byref(ap), ptr(out), _lib.current_stream(dev)),
                  'adfp_sample_tsdf')
        return out

    # ---- one sub-network alone (MLP.forward / mlp_tsdf.forward, reference decoder.py:177-203, :240-258) --------------
    def decode_single(self, mlp, pts, c, bound):
        """pts [P,3] f64/f32 -> [P] (low / high decoder) or [P,4] (colour decoder)."""
        _lib.require_cuda(pts, 'points')
        dev = pts.device
        name = mlp.name
        use_h = math_mode() == 'f16x3' and not mlp.__dict__.get('_single_exact')
        with _lib.device_guard(dev):
            mode = _lib.PTS_F64 if pts.dtype == torch.float64 else _lib.PTS_F32
            if mode == _lib.PTS_F32 and pts.dtype != torch.float32:
                pts = pts.float()
            pts = pts.detach().contiguous()
            P = pts.shape[0]
            _lib.check_status()
            sc = _lib.AdfpScene()
            sc.status = _lib.status_word().data_ptr()
            _lib.fill_bound(sc.bound, self.host_bound(bound, 'bound.' + name))
            packed = mlp._single_packed(name, 'h' if use_h else 'f32')
            setattr(sc, ('h_' if use_h else 'w_') + name, packed.data_ptr())
            keep = []
            for field, key in [(name, 'grid_' + name)] + ([('low', 'grid_low')] if name == 'high' else []):
                g = self.grid_cl(key, c[key])
                keep.append(g)
                gd = getattr(sc, field)
                gd.data, gd.Z, gd.Y, gd.X = g.data_ptr(), g.shape[0], g.shape[1], g.shape[2]
            ap = _lib.AdfpPoints()
            ap.mode, ap.n_points, ap.pts = mode, P, pts.data_ptr()
            out = torch.empty((P,) if name == 'high' else (P, 4), dtype=torch.float32, device=dev)
            if P:
                check(lib().adfp_decode_single(C.byref(sc), C.byref(ap), _lib.DEC_KIND[name], ptr(out), _lib.current_stream(dev)),
                      'adfp_decode_single')
                # a sub-network on its own is a convenience entry without the device-side repair: look at the range word now
                # and, if it tripped, evaluate again on the exact kernel (and stay there)
                if use_h and _lib.check_status(sync=True, device=dev) & _lib.STATUS_F16_RANGE:
                    mlp.__dict__['_single_exact'] = True
                    return self.decode_single(mlp, pts, c, bound)
            return out[:, 3].contiguous() if name == 'low' else out

    def attention_rows(self, mlp, pts, occ, tsdf_volume, tsdf_bnds):
        """(fused occupancy [M], attention weight [M]) of mlp_tsdf.forward at the points pts [M,3] with occupancies occ [M]."""
        _lib.require_cuda(pts, 'points')
        dev = pts.device
        use_h = math_mode() == 'f16x3' and not mlp.__dict__.get('_single_exact')
        tv = self.sample_tsdf(pts, tsdf_volume, tsdf_bnds)
        with _lib.device_guard(dev):
            M = tv.shape[0]
            occ = occ.detach().to(dev, torch.float32).contiguous()
            if occ.shape[0] != M:
                raise RuntimeError(f'mlp_tsdf: {occ.shape[0]} occupancies for {M} points')
            _lib.check_status()
            sc = _lib.AdfpScene()
            sc.status = _lib.status_word().data_ptr()
            packed = mlp._single_packed('att', 'h' if use_h else 'f32')
            setattr(sc, 'h_att' if use_h else 'w_att', packed.data_ptr())
            out4 = torch.empty((M, 4), dtype=torch.float32, device=dev)
            w = torch.empty((M,), dtype=torch.float32, device=dev)
            u = torch.empty((M,), dtype=torch.float32, device=dev)
            if M:
                check(lib().adfp_attention_rows(C.byref(sc), ptr(occ), ptr(tv), M, ptr(out4), ptr(w), ptr(u), _lib.current_stream(dev)),
                      'adfp_attention_rows')
                if use_h and _lib.check_status(sync=True, device=dev) & _lib.STATUS_F16_RANGE:
                    mlp.__dict__['_single_exact'] = True
                    return self.attention_rows(mlp, pts, occ, tsdf_volume, tsdf_bnds)
            return out4[:, 3].contiguous(), w

    # ---- a4..a13 ---------------------------------------------------------------------------
    def render_forward(self, decoders, c, rays_o, rays_d, gt_depth, tsdf_volume, tsdf_bnds, bound, stage,
                       n_samples, n_surface, lindisp=False, perturb=0.0, t_rand=None, depth_max=None,
                       want_aux=False, train=False, need_flat=None, depth_max_segment=0, depth_max_first_ray=0, tsdf_blocks=False,
                       frame=None, prefilter=None):
        """frame (not with train): dict(c2w=, H=, W=, fx=, fy=, cx=, cy=, depth=<the WHOLE frame's sensor depth>, n_rays=) -- the call
        renders pixels [depth_max_first_ray, + n_rays) of that frame (adfp_frame_job): rays_o / rays_d / gt_depth are ignored (pass
        None), the rays and the per-segment far clamps of the whole frame come out of the call's first launch.
        prefilter: (bound_dev [6] float64 device tensor, keep [N] uint8 device tensor) -- the Mapper's bounding-box pre-filter as a job of
        the call's first launch (adfp_render_args.prefilter_bound): keep is written, the far clamp is the kept rays' max depth."""
        f32 = torch.float32
        if frame is not None:
            fdepth = frame['depth'].detach().reshape(-1)
            _lib.require_cuda(fdepth, 'gt_depth')
            dev = fdepth.device
        else:
            _lib.require_cuda(rays_o, 'rays_o')
            dev = rays_o.device
        with _lib.device_guard(dev):
            if frame is not None:
                if fdepth.dtype != f32 or not fdepth.is_contiguous():
                    fdepth = fdepth.float().contiguous()
                N = int(frame['n_rays'])
                if fdepth.numel() != int(frame['H']) * int(frame['W']) or not (0 <= depth_max_first_ray and depth_max_first_ray + N <= fdepth.numel()):
                    raise RuntimeError(f"frame job: gt_depth has {fdepth.numel()} pixels for a {frame['H']}x{frame['W']} frame, pixels "
                                       f'[{depth_max_first_ray}, {depth_max_first_ray + N}) asked for')
                fc2w = frame['c2w'].detach().to(dev, f32).contiguous()
                if fc2w.numel() < 12:
                    raise RuntimeError(f'c2w: expected [3,4] or [4,4], got {tuple(fc2w.shape)}')
                ro = torch.empty((N, 3), dtype=f32, device=dev)
                rd = torch.empty((N, 3), dtype=f32, device=dev)
                gd = fdepth[depth_max_first_ray:depth_max_first_ray + N]
            else:
                ro = rays_o.detach()
                if ro.dtype != f32 or not ro.is_contiguous():
                    ro = ro.float().contiguous()
                rd = rays_d.detach()
                if rd.dtype != f32 or not rd.is_contiguous():
                    rd = rd.float().contiguous()
                N = ro.shape[0]
                gd = None
                if gt_depth is not None:
                    gd = gt_depth.detach().reshape(-1)
                    if gd.dtype != f32 or not gd.is_contiguous():
                        gd = gd.float().contiguous()
            S = n_samples + (n_surface if gd is not None else 0)
            depth = torch.empty((N,), dtype=torch.float64, device=dev)
            unc = torch.empty((N,), dtype=torch.float64, device=dev)
            color = torch.empty((N, 3), dtype=f32, device=dev)
            weight = torch.empty((N, S, 1), dtype=f32, device=dev)
            aux = None
            if N == 0:
                return depth, unc, color, weight, aux
            keys = {}
            if train:
                # buffers the backward reads after this call returns (never the shared workspace): one slab
                P = N * S
                extra = (('z_vals', 8 * P), ('raw', 16 * P))
                if self.export_relu_masks:
                    extra += tuple(('dbg_masks_' + n, 4 * (_lib.TRAIN_ATT_MASK_WORDS if n == 'att' else _lib.TRAIN_MASK_WORDS) * P)
                                   for n in _STAGE_NETS[stage])
                decoders.absorb_status()          # BEFORE the state is laid out (see eval_points_forward)
                aux = self.train_state(P, stage, dev, decoders, need_flat, extra=extra)
            sc, keep = self.scene(decoders, c, tsdf_volume, tsdf_bnds, bound, stage, keys=keys, state=aux, tsdf_blocks=tsdf_blocks,
                                  hand_over_packs=True)
            owed, self._owed_packs = self._owed_packs, None
            owed_rl, self._owed_relayouts = self._owed_relayouts, None
            launched = False
            try:
                a = _lib.AdfpRenderArgs()
                if owed is not None:                     # this call's first launch packs them beside its zero fill
                    a.pack_jobs, a.n_pack_jobs = C.cast(owed[0], C.c_void_p), owed[1]
                if owed_rl is not None:                  # ... and converts the grids
                    a.relayout_jobs, a.n_relayout_jobs = C.cast(owed_rl[0], C.c_void_p), owed_rl[1]
                a.stage = _lib.STAGE[stage]
                a.n_rays = N
                a.n_samples = n_samples
                a.n_surface = n_surface
                a.lindisp = 1 if lindisp else 0
                a.perturb = float(perturb)
                a.rays_o = ro.data_ptr()
                a.rays_d = rd.data_ptr()
                a.gt_depth = gd.data_ptr() if gd is not None else None
                if perturb > 0:
                    t_rand = t_rand.to(dev, f32).contiguous()
                    a.t_rand = t_rand.data_ptr()
                a.depth_max_segment = int(depth_max_segment)
                a.depth_max_first_ray = int(depth_max_first_ray)
                if prefilter is not None:
                    a.prefilter_bound, a.prefilter_keep = prefilter[0].data_ptr(), prefilter[1].data_ptr()
                if frame is not None:
                    fj = _lib.AdfpFrameJob()
                    fj.c2w, fj.H, fj.W = fc2w.data_ptr(), int(frame['H']), int(frame['W'])
                    fj.fx, fj.fy, fj.cx, fj.cy = float(frame['fx']), float(frame['fy']), float(frame['cx']), float(frame['cy'])
                    fj.depth, fj.rays_o, fj.rays_d = fdepth.data_ptr(), ro.data_ptr(), rd.data_ptr()
                    a.frame = C.pointer(fj)
                if depth_max is not None:
                    depth_max = depth_max.to(dev, f32).reshape(-1).contiguous()
                    a.depth_max = depth_max.data_ptr()
                a.depth = depth.data_ptr()
                a.uncertainty = unc.data_ptr()
                a.color = color.data_ptr()
                a.weight = weight.data_ptr()
                if train:
                    a.z_vals, a.raw = aux['ptrs']['z_vals'], aux['ptrs']['raw']
                    aux.update(rays_o=ro, rays_d=rd, S=S, N=N, keys=keys)
                    a.state = C.pointer(aux['_state'])
                    if want_aux:
                        aux['z_vals'] = self.state_tensor(aux, 'z_vals', torch.float64, (N, S))
                        aux['raw'] = self.state_tensor(aux, 'raw', f32, (N, S, 4))
                elif want_aux:
                    aux = {'z_vals': torch.empty((N, S), dtype=torch.float64, device=dev),
                           'raw': torch.empty((N, S, 4), dtype=f32, device=dev)}
                    a.z_vals = aux['z_vals'].data_ptr()
                    a.raw = aux['raw'].data_ptr()
                ws = self.workspace(N * S, dev)
                a.workspace = ws.data_ptr()
                a.workspace_bytes = ws.numel()
                check(lib().adfp_render_forward(C.byref(sc), C.byref(a), _lib.current_stream(dev)), 'adfp_render_forward')
                launched = True
            finally:
                if owed is not None and not launched:      # the images the cache already calls current were never packed
                    decoders._packed = {}
                if owed_rl is not None and not launched:   # nor the channels-last copies written
                    self._grid_cache = {k: v for k, v in self._grid_cache.items() if len(v) == 4}
        return depth, unc, color, weight, aux

    # ---- a15 -------------------------------------------------------------------------------
    def render_backward(self, decoders, c, tsdf_volume, tsdf_bnds, bound, stage, saved, g_depth, g_unc, g_color,
                        g_weight, need_grid, need_flat, need_rays=False, ray_keep=None, out_grids=None, out_flats=None,
                        out_grids_cl=None, grids_prezeroed=False, side_lane=False):
        """saved: the aux dict of render_forward(train=True).  need_grid / need_flat: dicts of bools.
        out_grids / out_flats: optional caller-owned result tensors (name -> [1,32,Z,Y,X] / flat), e.g. slices of one
        gradient bucket that is all-reduced as it stands.
        out_grids_cl: caller-owned CHANNELS-LAST gradient tensors (name -> [Z,Y,X,32]): the kernels' own layout, handed back as
        it is (no re-layout); grids_prezeroed = they are all zero on entry (mapping.MapperIteration keeps them so).
        Returns (grid grads dict in the reference's [1,32,Z,Y,X] layout -- or channels-last for out_grids_cl --, flat parameter
        grads dict).
        side_lane: run the spatial sort on the engine's second stream (side_lane()).  For callers whose iteration is bound by the
        GPU (mapping.MapperIteration: one graph replay per iteration); the autograd path of the unchanged callers is bound by host
        time, where the lane's four event calls cost more than the overlap returns (profiles/r06_host_breakdown.txt)."""
        ro = saved['rays_o']
        dev = ro.device
        L = lib()
        f32 = torch.float32
        with _lib.device_guard(dev):
            N, S = ro.shape[0], saved['S']
            ht = self.ht_nets(saved, need_flat, need_rays, need_grid)
            saved['ht_used'] = ht
            sc, keep = self.scene(decoders, c, tsdf_volume, tsdf_bnds, bound, stage, backward=True, ht_nets=ht, keys=saved.get('keys'))
            a = _lib.AdfpBackwardArgs()
            a.stage = _lib.STAGE[stage]
            a.n_rays, a.S = N, S
            a.rays_o, a.rays_d = ro.data_ptr(), saved['rays_d'].data_ptr()
            a.z_vals, a.raw = saved['ptrs']['z_vals'], saved['ptrs']['raw']
            a.state = saved['_state']

            def prep(t, dtype):
                if t is None:
                    return None
                t = t.detach()
                if t.dtype != dtype or t.device != dev:
                    t = t.to(dev, dtype)
                return t if t.is_contiguous() else t.contiguous()
            gd, gu, gc, gw = prep(g_depth, torch.float64), prep(g_unc, torch.float64), prep(g_color, f32), prep(g_weight, f32)
            a.g_depth, a.g_uncertainty = _lib.ptr(gd), _lib.ptr(gu)
            a.g_color, a.g_weight = _lib.ptr(gc), _lib.ptr(gw)
            if ray_keep is not None:
                a.ray_keep = ray_keep.data_ptr()
            grids_cl, flats = {}, {}
            direct = {}
            # Gradient outputs the caller did not provide come out of ONE allocation (the three grids first: their channel-major
            # copies are written with 16-byte stores; then the flat parameter gradients): one torch.empty instead of five, and
            # dist.allreduce_grads finds the whole bucket contiguous -- the .grad tensors autograd keeps are views of it -- and
            # all-reduces it in place, with no packing copy either way.
            own_grid = [(name, key) for name, key in (('low', 'grid_low'), ('high', 'grid_high'), ('color', 'grid_color'))
                        if need_grid.get(name) and not (out_grids_cl and name in out_grids_cl) and not (out_grids and name in out_grids)]
            own_flat = [name for name in ('low', 'high', 'color', 'att') if need_flat.get(name) and not (out_flats and name in out_flats)]
            bucket, boff = None, {}
            if own_grid or own_flat:
                off = 0
                for name, key in own_grid:
                    boff['g' + name] = off
                    off += c[key].numel()
                for name in own_flat:
                    boff['f' + name] = off
                    off += _flat_floats(name)
                bucket = torch.empty((off,), dtype=f32, device=dev)
            for name, key in (('low', 'grid_low'), ('high', 'grid_high'), ('color', 'grid_color')):
                if need_grid.get(name):
                    Z, Y, X = c[key].shape[2:]
                    if out_grids_cl and name in out_grids_cl:
                        direct[name] = out_grids_cl[name]
                        setattr(a, 'g_grid_' + name, direct[name].data_ptr())
                        continue
                    grids_cl[name] = self._cl_scratch(name, (Z, Y, X, 32), dev)
                    setattr(a, 'g_grid_' + name, grids_cl[name].data_ptr())
            for name in ('low', 'high', 'color', 'att'):
                if need_flat.get(name):
                    if out_flats and name in out_flats:
                        flats[name] = out_flats[name]
                    else:
                        o = boff['f' + name]
                        flats[name] = bucket[o:o + _flat_floats(name)]
                    setattr(a, 'g_flat_' + name, flats[name].data_ptr())
            g_ro = g_rd = None
            if need_rays:
                g_ro = torch.empty((N, 3), dtype=f32, device=dev)
                g_rd = torch.empty((N, 3), dtype=f32, device=dev)
                a.g_rays_o, a.g_rays_d = g_ro.data_ptr(), g_rd.data_ptr()
            ws = self.bwd_workspace(N * S, dev)
            a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
            a.options = self.bwd_options | (_lib.BWD_GRIDS_PREZEROED if (grids_prezeroed and not grids_cl) else 0)
            lane = self.side_lane(dev) if side_lane else None
            if lane is not None:
                a.side_stream = lane[0].cuda_stream
                for k in range(2):
                    a.side_events[k] = lane[1][k].cuda_event
            stream = _lib.current_stream(dev)
            check(L.adfp_render_backward(C.byref(sc), C.byref(a), stream), 'adfp_render_backward')
            grids = dict(direct)
            if grids_cl:                                  # the kernels' channels-last gradients -> the reference's layout: ONE launch
                arr = (_lib.AdfpRelayoutJob * len(grids_cl))()
                for k, (name, g) in enumerate(grids_cl.items()):
                    Z, Y, X = g.shape[:3]
                    if out_grids and name in out_grids:
                        out = out_grids[name]
                    else:
                        o = boff['g' + name]
                        out = bucket[o:o + 32 * Z * Y * X].view(1, 32, Z, Y, X)
                    arr[k].src, arr[k].dst, arr[k].voxels = g.data_ptr(), out.data_ptr(), Z * Y * X
                    grids[name] = out
                check(L.adfp_relayout_grids(len(grids_cl), arr, 1, stream), 'adfp_relayout_grids')
        return grids, flats, (g_ro, g_rd)
