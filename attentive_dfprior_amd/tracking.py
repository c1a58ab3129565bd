"""
The Tracker's iteration as ONE fixed kernel sequence (reference src/Tracker.py:75-134, Tracker.optimize_cam_in_batch, and the
loop around it, :236-263), replayable from a HIP graph -- the camera-tracking counterpart of mapping.MapperIteration.

The reference-shaped way (common.get_samples -> common.filter_rays_in_bound -> Renderer.render_batch_ray -> torch loss ->
loss.backward() -> torch.optim.Adam) works against this package too and is what tests compare with; at the Tracker's 200 - 1000
rays it is bound by host dispatch (about forty small launches, two host read-backs and the autograd engine per iteration), not
by the GPU.  Here the iteration is

    adfp_camera_from_tensor          quaternion + translation -> c2w                                   (src/common.py:139-178)
    adfp_select_pixels               the torch.randint draw -> pixel coordinates, sensor depth, colour  (src/common.py:94-124)
    adfp_rays_from_uv                rays                                                               (src/common.py:76-91)
    adfp_prefilter_mask              the bounding-box pre-filter as a keep flag + the kept rays' far clamp   (Tracker.py:100-109)
    adfp_render_forward              stage color, training state kept                                   (Tracker.py:111-113)
    adfp_tracker_loss                loss, outlier mask (median), cotangents                            (Tracker.py:115-129)
    adfp_render_backward             -> d loss / d rays_o, rays_d only (decoders and grids are frozen)
    adfp_rays_from_uv_backward       -> d loss / d c2w
    adfp_camera_from_tensor_backward -> d loss / d camera tensor
    adfp_adam_prep + adfp_masked_adam_multi      torch.optim.Adam on the 7 pose parameters             (Tracker.py:131-133)
    adfp_track_keep_best             candidate_cam_tensor                                              (Tracker.py:261-263)

with no host read-back anywhere: dropped rays are rendered and get no gradient (a kept ray's result does not depend on its
neighbours -- the far clamp is taken over the kept rays only), the loss stays on the device.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, ptr, check


class TrackerIteration(object):
    """
    it = TrackerIteration(renderer, decoders, c, tsdf_volume, tsdf_bnds, H, W, fx, fy, cx, cy, ignore_edge_h, ignore_edge_w)
    for every frame:
        it.new_frame(camera_tensor, gt_depth, gt_color)           # fresh Adam, like the reference (src/Tracker.py:219-229)
        for cam_iter in range(num_cam_iters):
            loss = it.step(tracking_pixels)                       # device float64, no sync
        c2w = common.get_camera_from_tensor(it.best_camera_tensor)

    cam_lr, seperate_LR      tracking.lr / tracking.seperate_LR (sic) of the reference's config: with seperate_LR the translation steps
                             with cam_lr and the quaternion with 0.2 cam_lr (:225-226)
    use_color, w_color_loss  tracking.use_color_in_tracking / tracking.w_color_loss
    handle_dynamic           tracking.handle_dynamic
    """

    def __init__(self, renderer, decoders, c, tsdf_volume, tsdf_bnds, H, W, fx, fy, cx, cy, ignore_edge_h=0, ignore_edge_w=0,
                 cam_lr=1e-3, seperate_LR=False, use_color=True, w_color_loss=0.5, handle_dynamic=True,
                 betas=(0.9, 0.999), eps=1e-8, use_graph=True):
        self.rend, self.dec, self.c = renderer, decoders, c
        self.tsdf, self.tsdf_bnds = tsdf_volume, tsdf_bnds
        self.H, self.W, self.intr = int(H), int(W), (float(fx), float(fy), float(cx), float(cy))
        self.window = (int(ignore_edge_h), int(H) - int(ignore_edge_h), int(ignore_edge_w), int(W) - int(ignore_edge_w))
        self.cam_lr, self.separate = float(cam_lr), bool(seperate_LR)
        self.w_color = float(w_color_loss) if use_color else 0.0
        self.handle_dynamic = bool(handle_dynamic)
        self.betas, self.eps, self.use_graph = betas, eps, use_graph
        self.dev = dev = next(iter(c.values())).device
        for k, g in c.items():
            _lib.require_cuda(g, k)
        f32 = dict(dtype=torch.float32, device=dev)
        self.cam = torch.zeros(7, **f32)                       # the pose being optimised: quaternion (r, i, j, k) + translation
        self.best_cam = torch.zeros(7, **f32)
        self.best_loss = torch.full((1,), float('inf'), dtype=torch.float64, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float64, device=dev)
        self.exp_avg, self.exp_avg_sq, self.g_cam = torch.zeros(7, **f32), torch.zeros(7, **f32), torch.zeros(7, **f32)
        self.c2w, self.g_c2w = torch.zeros(16, **f32), torch.zeros(16, **f32)
        self.depth_img = torch.zeros((self.H, self.W), **f32)
        self.color_img = torch.zeros((self.H, self.W, 3), **f32)
        # Adam groups as torch.optim.Adam sees them: [T, quaternion] with seperate_LR, else the one 7-vector
        self.n_groups = 2 if self.separate else 1
        self.step_count = torch.zeros(self.n_groups, dtype=torch.int32, device=dev)
        self.derived = torch.empty((self.n_groups, 2), **f32)
        self.bound_dev = torch.as_tensor(renderer.bound).to(dev, torch.float64).contiguous()
        self._graphs, self._pick, self._pool = {}, {}, None
        # The captured graphs read the grids through channels-last copies.  The iteration OWNS those copies (one per grid, re-used
        # in place when a grid of the same shape is handed over) and registers them with the engine's layout cache before every
        # eager pass and every replay: the cache has one slot per grid name and is shared with every other user of the same
        # Renderer (a Mapper in the same process replaces or clears the slot), so a pointer baked into a graph must never be the
        # cache's own buffer.
        self._shadow, self._shadow_src = {}, {}

    # ---- per frame ----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def new_frame(self, camera_tensor, gt_depth, gt_color):
        """camera_tensor [7] (common.get_tensor_from_camera of the initial guess); gt_depth [H,W], gt_color [H,W,3].  Everything is
        copied into buffers the captured graphs read, so those stay valid from frame to frame."""
        self.cam.copy_(camera_tensor.detach().reshape(7))
        self.best_cam.copy_(self.cam)
        self.best_loss.fill_(float('inf'))
        self.depth_img.copy_(gt_depth)
        self.color_img.copy_(gt_color)
        for t in (self.exp_avg, self.exp_avg_sq):
            t.zero_()
        self.step_count.zero_()

    @property
    def camera_tensor(self):
        return self.cam

    @property
    def best_camera_tensor(self):
        """The pose of the lowest-loss iteration so far (candidate_cam_tensor, src/Tracker.py:261-263)."""
        return self.best_cam

    # ---- the kernel sequence ------------------------------------------------------------------------------------------
    def _sequence(self, pick, adam=True):
        L = lib()
        dev, eng, dec, rend = self.dev, self.rend._engine, self.dec, self.rend
        st = _lib.current_stream(dev)
        n = pick.shape[0]
        fx, fy, cx, cy = self.intr
        H0, H1, W0, W1 = self.window
        f32 = dict(dtype=torch.float32, device=dev)
        # camera tensor -> c2w, the drawn pixels, their rays, the bounding-box keep flags and the kept rays' largest depth: ONE launch
        # (adfp_tracker_head = adfp_camera_from_tensor + adfp_select_pixels + adfp_rays_from_uv + adfp_prefilter_mask)
        pi, pj, gd = torch.empty(n, **f32), torch.empty(n, **f32), torch.empty(n, **f32)
        gc = torch.empty((n, 3), **f32)
        ro, rd = torch.empty((n, 3), **f32), torch.empty((n, 3), **f32)
        keep = torch.empty((n,), dtype=torch.uint8, device=dev)
        dmax = torch.empty((1,), **f32)
        ha = _lib.AdfpTrackerHeadArgs()
        ha.cam, ha.c2w, ha.idx, ha.n = self.cam.data_ptr(), self.c2w.data_ptr(), pick.data_ptr(), n
        ha.H0, ha.H1, ha.W0, ha.W1, ha.H, ha.W = H0, H1, W0, W1, self.H, self.W
        ha.depth_img, ha.color_img = self.depth_img.data_ptr(), self.color_img.data_ptr()
        ha.fx, ha.fy, ha.cx, ha.cy, ha.bound = fx, fy, cx, cy, self.bound_dev.data_ptr()
        ha.pix_i, ha.pix_j, ha.gt_depth, ha.gt_color = pi.data_ptr(), pj.data_ptr(), gd.data_ptr(), gc.data_ptr()
        ha.rays_o, ha.rays_d, ha.keep, ha.depth_max = ro.data_ptr(), rd.data_ptr(), keep.data_ptr(), dmax.data_ptr()
        check(L.adfp_tracker_head(C.byref(ha), st), 'adfp_tracker_head')
        no_flat = {k: False for k in ('low', 'high', 'color', 'att')}
        depth, unc, color, weight, aux = eng.render_forward(dec, self.c, ro, rd, gd, self.tsdf, self.tsdf_bnds, rend.bound, 'color',
                                                            rend.N_samples, rend.N_surface, rend.lindisp, rend.perturb, None, dmax,
                                                            train=True, need_flat=no_flat)
        la = _lib.AdfpTrackLossArgs()
        la.n_rays, la.handle_dynamic, la.w_color_loss = n, 1 if self.handle_dynamic else 0, self.w_color
        la.depth, la.uncertainty, la.color = depth.data_ptr(), unc.data_ptr(), color.data_ptr()
        la.gt_depth, la.gt_color, la.keep = gd.data_ptr(), gc.data_ptr(), keep.data_ptr()
        g_depth = torch.empty((n,), dtype=torch.float64, device=dev)
        g_color = torch.empty((n, 3), **f32)
        la.loss, la.g_depth, la.g_color = self.loss.data_ptr(), g_depth.data_ptr(), g_color.data_ptr()
        check(L.adfp_tracker_loss(C.byref(la), st), 'adfp_tracker_loss')
        none = {k: False for k in ('low', 'high', 'color')}
        _, _, (g_ro, g_rd) = eng.render_backward(dec, self.c, self.tsdf, self.tsdf_bnds, rend.bound, 'color', aux, g_depth, None,
                                                 g_color, None, none, no_flat, need_rays=True, ray_keep=keep)
        # ray cotangents -> d/d c2w -> d/d camera tensor, and (adam) the Adam step on the pose's parameter groups and the running best
        # pose: ONE launch (adfp_tracker_tail = adfp_rays_from_uv_backward + adfp_camera_from_tensor_backward + adfp_adam_prep +
        # adfp_masked_adam_multi + adfp_track_keep_best).  The candidate is kept BEFORE the step with separate learning rates
        # (camera_tensor = cat([quad, T]) is rebuilt before the step in the reference loop, :237-238, so the pose it clones after the step
        # is the one this loss was measured at) and after it with the one-tensor optimiser (updated in place).
        ta = _lib.AdfpTrackerTailArgs()
        ta.pix_i, ta.pix_j, ta.n, ta.fx, ta.fy, ta.cx, ta.cy = pi.data_ptr(), pj.data_ptr(), n, fx, fy, cx, cy
        ta.g_rays_o, ta.g_rays_d = g_ro.data_ptr(), g_rd.data_ptr()
        ta.cam, ta.g_c2w, ta.g_cam = self.cam.data_ptr(), self.g_c2w.data_ptr(), self.g_cam.data_ptr()
        ta.step = 1 if adam else 0
        if adam:
            ta.exp_avg, ta.exp_avg_sq = self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
            ta.steps, ta.derived, ta.n_groups = self.step_count.data_ptr(), self.derived.data_ptr(), self.n_groups
            ta.lr[0], ta.lr[1] = (self.cam_lr, self.cam_lr * 0.2) if self.separate else (self.cam_lr, 0.0)
            ta.beta1, ta.beta2, ta.eps = self.betas[0], self.betas[1], self.eps
            # a forward repaired for an f16-range event has zero gradients by construction: nobody steps then (adfp_train_state.counter[8])
            ta.skip_flag = aux['counter_ptr'] + 32
            ta.loss, ta.best_loss, ta.best_cam = self.loss.data_ptr(), self.best_loss.data_ptr(), self.best_cam.data_ptr()
        check(L.adfp_tracker_tail(C.byref(ta), st), 'adfp_tracker_tail')

    @torch.no_grad()
    def gradient(self, batch_size, pick=None):
        """d loss / d camera tensor at the current pose without stepping (tests, diagnostics): (loss, g [7]) device tensors."""
        with torch.cuda.device(self.dev):
            self._sync_shadows()
            self._sequence(self._draw(batch_size) if pick is None else pick.to(self.dev, torch.int64).contiguous(), adam=False)
        return self.loss.clone(), self.g_cam.clone()

    def _sync_shadows(self):
        """Every grid's channels-last shadow holds the grid as it stands (re-laid out here, outside any graph, when the grid was
        replaced or written since) and is what Engine.scene() finds for it."""
        eng = self.rend._engine
        for k, g in self.c.items():
            _lib.require_cuda(g, k)
            src = (g.data_ptr(), g._version, g.shape, g.stride())
            sh = self._shadow.get(k)
            Z, Y, X = g.shape[2:]
            if sh is None or sh.shape != (Z, Y, X, 32) or sh.device != g.device:
                sh = self._shadow[k] = torch.empty((Z, Y, X, 32), dtype=torch.float32, device=g.device)
                self._shadow_src[k] = None
            held = self._shadow_src[k]
            if held is None or held[0] != src:
                gc = g.detach()
                if gc.dtype != torch.float32 or not gc.is_contiguous():
                    gc = gc.float().contiguous()
                check(lib().adfp_relayout_grid(ptr(gc), ptr(sh), 32, Z, Y, X, _lib.current_stream(g.device)), 'adfp_relayout_grid')
                # the entry keeps the source's STORAGE alive: (data_ptr, _version) names the contents only as long as the allocator
                # cannot hand the same address, at version 0, to the next grid (a grid replaced twice between two steps)
                self._shadow_src[k] = (src, g.untyped_storage())
            eng.adopt_grid_cl(k, g, sh)

    def _draw(self, n):
        H0, H1, W0, W1 = self.window
        return torch.randint((H1 - H0) * (W1 - W0), (n,), device=self.dev)         # the reference's RNG call (src/common.py:101)

    @torch.no_grad()
    def step(self, batch_size, pick=None):
        """One iteration on `batch_size` random pixels (or the given window indices `pick`, int64); returns the loss as a device
        float64 tensor -- reading it synchronises; the reference reads it every iteration (loss.item(), :134), the fused loop does
        not need to: the best pose is tracked on the device."""
        dev = self.dev
        with torch.cuda.device(dev):
            n = int(batch_size)
            if pick is None:
                pick = self._draw(n)
            self._sync_shadows()                            # before the eager sequence, the warm-up, the capture AND every replay
            if not self.use_graph:
                self._sequence(pick.to(dev, torch.int64).contiguous())
                return self.loss
            self.dec.absorb_status()                        # a replay never passes through Engine.scene(): see MapperIteration.step

            def graph_key():
                return (n, frozenset(self.dec._exact_latch), self._scene_token())
            key = graph_key()
            buf = self._pick.get(n)
            if buf is None:
                buf = self._pick[n] = torch.empty((n,), dtype=torch.int64, device=dev)
            buf.copy_(pick)
            g = self._graphs.get(key)
            if g is None:
                # warm the host-side caches (packed weight images, channels-last grids, bounds) eagerly, without stepping: the Tracker's
                # networks and grids are frozen, so everything the caches hold stays valid and none of it is rebuilt in the graph
                for _ in range(2):                         # a second pass if the first one latched a network onto the exact kernels
                    side = torch.cuda.Stream(device=dev)
                    side.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(side):
                        self._sequence(buf, adam=False)
                    torch.cuda.current_stream(dev).wait_stream(side)
                    torch.cuda.synchronize(dev)
                    self.dec.absorb_status()
                    if graph_key() == key:
                        break
                    key = graph_key()
                self._graphs = {k: v for k, v in self._graphs.items() if k[2] == key[2]}      # graphs of replaced scene tensors are dead
                eng = self.rend._engine
                g = torch.cuda.CUDAGraph()
                if self._pool is None:
                    self._pool = torch.cuda.graph_pool_handle()
                with eng.private_workspaces():              # the graph owns its workspaces
                    with torch.cuda.graph(g, pool=self._pool):
                        self._sequence(buf)
                self._graphs[key] = g
            g.replay()
            return self.loss

    def _scene_token(self):
        """Identity of everything a captured graph has baked in: the iteration's own channels-last grid copies (their ADDRESSES --
        the contents are refreshed in place by _sync_shadows, so new grids of the same shape replay the same graph), the decoder
        parameters (identity + version: the packed weight images are made from them outside the graph, and re-made in place)
        and the TSDF volume."""
        ts = list(self.dec.parameters()) + [self.tsdf]
        return hash((tuple(sh.data_ptr() for sh in self._shadow.values()), tuple((t.data_ptr(), t._version) for t in ts)))

    def update_para(self, decoders=None, c=None):
        """Tracker.update_para_from_mapping (src/Tracker.py:136-147) hands over fresh copies of the decoders and grids; graphs
        captured against the old ones are dropped at the next step (they are keyed on the tensors' identity and version)."""
        if decoders is not None:
            self.dec = decoders
        if c is not None:
            self.c = c
            # re-lay the handed-over grids out at the next step WHATEVER their address and version say: a writer that goes through raw
            # pointers (another process's kernels on IPC memory) does not bump this process's version counters.  External writers
            # that keep writing a grid the iteration holds must call this again (or bump the version).
            self._shadow_src = {k: None for k in self._shadow_src}
