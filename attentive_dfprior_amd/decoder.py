"""
Drop-in for the reference's ``src/conv_onet/models/decoder.py``: the same nn.Module tree
(``DF`` -> ``low_decoder`` / ``high_decoder`` / ``color_decoder`` : ``MLP``, ``mlp`` : ``mlp_tsdf``)
with the same parameter names (``fc_c.{i}``, ``embedder._B``, ``pts_linears.{i}``,
``output_linear``), so ``state_dict`` / ``load_state_dict`` / ``deepcopy`` / ``share_memory`` /
``.parameters()`` behave as src/DF_Prior.py:191-218, src/Mapper.py:364-371, src/Tracker.py:144
and src/utils/Logger.py:26 expect -- but ``DF.forward`` runs on the MI355X through libadfp.so.

The modules only OWN parameters; all arithmetic happens in the HIP kernels.  There is no
PyTorch fallback: calling ``DF`` on CPU tensors raises.
"""

import torch
import torch.nn as nn

from . import _lib
from ._lib import lib


class GaussianFourierFeatureTransform(nn.Module):
    """Owner of the [3, 93] Gaussian projection ``_B`` (reference decoder.py:7-30)."""

    def __init__(self, num_input_channels, mapping_size=93, scale=25, learnable=True):
        super().__init__()
        b = torch.randn((num_input_channels, mapping_size)) * scale
        if learnable:
            self._B = nn.Parameter(b)
        else:
            self._B = b


class DenseLayer(nn.Linear):
    """nn.Linear with xavier-uniform(gain(activation)) weights and zero bias (decoder.py:68-77)."""

    def __init__(self, in_dim, out_dim, activation='relu', *args, **kwargs):
        self.activation = activation
        super().__init__(in_dim, out_dim, *args, **kwargs)

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.weight, gain=nn.init.calculate_gain(self.activation))
        if self.bias is not None:
            nn.init.zeros_(self.bias)


_PACK_SIZES = {}


def _pack_size(name, fmt):
    """Element count of a packed image (int32 words for the f16-split images, floats for the exact one); constants of the library."""
    hit = _PACK_SIZES.get((name, fmt))
    if hit is None:
        L = lib()
        f = 'h' if fmt in ('g', 'hg') else fmt                 # one buffer holds the H and the G layout (adfp_pack_split_image)
        if name == 'att':
            hit = {'h': L.adfp_attention_packed_h_words, 'ht': L.adfp_attention_packed_ht_words, 'f32': L.adfp_attention_packed_floats}[f]()
        else:
            kind = _lib.DEC_KIND[name]
            hit = {'h': L.adfp_decoder_packed_h_words, 'ht': L.adfp_decoder_packed_ht_words, 'f32': L.adfp_decoder_packed_floats}[f](kind)
        _PACK_SIZES[(name, fmt)] = hit = int(hit)
    return hit


def pack_jobs_array(jobs):
    """The ctypes form of a list of deferred pack jobs: (array, keep-alive list)."""
    arr = (_lib.AdfpPackJob * max(1, len(jobs)))()
    for k, (net, fmt, flat, packed) in enumerate(jobs):
        arr[k].net, arr[k].format, arr[k].flat, arr[k].packed = net, fmt, flat.data_ptr(), packed.data_ptr()
    return arr, [t for j in jobs for t in j[2:]]


def flush_pack_jobs(jobs, status, device):
    """ONE launch for the deferred split-image / transposed-image packs of a call (pack_network(..., defer=jobs)): a pack kernel
    costs ~5 us inside a graph replay whatever it packs, and a training iteration re-packs four images per step."""
    if not jobs:
        return
    arr, _ = pack_jobs_array(jobs)
    sptr = _lib.status_ptr() if status is None else _lib.C.c_void_p(status.data_ptr())
    with _lib.device_guard(device):
        _lib.check(lib().adfp_pack_images(len(jobs), arr, sptr, _lib.current_stream(device)), 'adfp_pack_images')
    del jobs[:]


def pack_network(name, params, fmt='f32', status=None, flat=None, out=None, defer=None):
    """Packed image of one sub-network ('low' / 'high' / 'color' / 'att') from its parameters in state_dict order.
    fmt: 'f32' (adfp_pack_decoder / adfp_pack_attention), 'ht' (adfp_pack_*_ht); 'h' / 'g' / 'hg': the split image with its H
    (32x32x16 order: training forward, single-network entries) / G (16x16x32 order: inference kernels) / both parts written.
    status: the pinned status word a weight outside the f16 range is reported to (default: the process-wide one).
    out: an image of the same network and format to overwrite (a training iteration re-packs the trained networks every step:
    same-stream ordering makes the in-place rebuild safe, and the image keeps its address).
    defer: a list -- the split-image parts and the transposed image are not packed here but appended as jobs
    (net, ADFP_IMAGE_*, flat, packed) for flush_pack_jobs; the caller flushes before it launches anything that reads an image."""
    if flat is None:
        flat = _flat_params(params)
    sptr = _lib.status_ptr() if status is None else _lib.C.c_void_p(status.data_ptr())
    _lib.require_cuda(flat, f'{name} decoder parameters')
    L = lib()
    dev = flat.device
    n = _pack_size(name, fmt)
    dtype = torch.float32 if fmt == 'f32' else torch.int32
    if out is not None and (out.numel() != n or out.dtype != dtype or out.device != dev):
        out = None
    with _lib.device_guard(dev):
        stream = _lib.current_stream(dev)
        packed = out if out is not None else torch.empty(n, dtype=dtype, device=dev)
        if defer is not None and fmt in ('h', 'g', 'hg', 'ht') and len(defer) + 2 <= 8:
            for part, bit in (('h', _lib.IMAGE_H), ('g', _lib.IMAGE_G), ('ht', _lib.IMAGE_HT)):
                if part == fmt or (fmt == 'hg' and part != 'ht'):
                    defer.append((_lib.NET_ID[name], bit, flat, packed))
            return packed
        if fmt in ('h', 'g', 'hg'):
            which = {'h': _lib.IMAGE_H, 'g': _lib.IMAGE_G, 'hg': _lib.IMAGE_H | _lib.IMAGE_G}[fmt]
            _lib.check(L.adfp_pack_split_image(_lib.NET_ID[name], which, _lib.ptr(flat), _lib.ptr(packed), sptr, stream), 'adfp_pack_split_image')
        elif fmt in ('h', 'ht'):
            if name == 'att':
                fn = L.adfp_pack_attention_h if fmt == 'h' else L.adfp_pack_attention_ht
                _lib.check(fn(_lib.ptr(flat), _lib.ptr(packed), sptr, stream), 'adfp_pack_attention_' + fmt)
            else:
                kind = _lib.DEC_KIND[name]
                fn = L.adfp_pack_decoder_h if fmt == 'h' else L.adfp_pack_decoder_ht
                _lib.check(fn(kind, _lib.ptr(flat), _lib.ptr(packed), sptr, stream), 'adfp_pack_decoder_' + fmt)
        elif name == 'att':
            assert flat.numel() == L.adfp_attention_flat_floats()
            _lib.check(L.adfp_pack_attention(_lib.ptr(flat), _lib.ptr(packed), stream), 'adfp_pack_attention')
        else:
            kind = _lib.DEC_KIND[name]
            assert flat.numel() == L.adfp_decoder_flat_floats(kind)
            _lib.check(L.adfp_pack_decoder(kind, _lib.ptr(flat), _lib.ptr(packed), stream), 'adfp_pack_decoder')
    return packed


class _SingleNet(object):
    """Mixin of MLP / mlp_tsdf: the packed image of THIS module alone, for direct calls of a sub-network
    (``decoders.low_decoder(p, c_grid)``, reference decoder.py:177).  The cache is dropped by deepcopy / pickling.

    WHERE the parameters live.  The kernels want a network's parameters as one flat float32 buffer in state_dict order.  They
    are put there at the moments the storage is replaced ANYWAY and nobody can hold the old one yet: ``.to(device)`` /
    ``.cuda()`` / ``.float()`` (``_apply``) and ``copy.deepcopy`` -- never lazily inside a render call.  The reference shares
    the decoders between its Mapper and Tracker processes through CUDA IPC after ``.to(device)`` (src/DF_Prior.py:50-51,
    :108-110, :305-307): the flat buffer then IS the exported storage, every nn.Parameter a view of it on both sides, and an
    in-place optimiser step in the Mapper process is what the Tracker process reads.  Parameters that are not in one buffer
    (somebody assigned ``p.data``) are served from a copy cached on their versions (DF.flat_weights)."""

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        home_parameters(self)
        return out

    def _single_packed(self, name, fmt):
        params = tuple(self.parameters())
        key = (_version_key(params), fmt)
        hit = self.__dict__.get('_single')
        if hit is not None and hit[0] == key and not self.__dict__.get('_foreign'):
            return hit[1]
        packed = pack_network(name, params, fmt)
        self.__dict__['_single'] = (key, packed)
        return packed

    def __getstate__(self):
        d = self.__dict__.copy()
        d.pop('_single', None)
        d.pop('_single_exact', None)
        return d

    def __setstate__(self, state):
        # unpickled = received from another process (torch.multiprocessing spawn, src/DF_Prior.py:302-311): see DF.__setstate__
        super().__setstate__(state)              # nn.Module's own: the dict update + its default-attribute fix-ups for older pickles
        self.__dict__['_foreign'] = True

    @staticmethod
    def _refuse_autograd(*tensors_and_modules):
        if not torch.is_grad_enabled():
            return
        for t in tensors_and_modules:
            ts = t.parameters() if isinstance(t, nn.Module) else (t.values() if isinstance(t, dict) else [t])
            if any(isinstance(x, torch.Tensor) and x.requires_grad for x in ts):
                raise NotImplementedError(
                    'a sub-network called on its own is inference-only in libadfp (use torch.no_grad()); '
                    'gradients flow through the DF module (DF.forward / Renderer.render_batch_ray / eval_points)')


_shared_engine = None


def _engine():
    global _shared_engine
    if _shared_engine is None:
        from .engine import Engine
        _shared_engine = Engine()
    return _shared_engine


class MLP(_SingleNet, nn.Module):
    """Parameter container of one conv_onet decoder (reference decoder.py:91-166).

    Fourier-93 -> 5 x 32 ReLU layers with per-layer feature injection ``fc_c`` and a skip
    concat after layer 2.  Only ``pos_embedding_method='fourier'`` (configs/df_prior.yaml:103)
    is implemented by the kernels.
    """

    def __init__(self, name='', dim=3, c_dim=128, hidden_size=256, n_blocks=5, leaky=False,
                 sample_mode='bilinear', color=False, skips=[2], grid_len=0.16,
                 pos_embedding_method='fourier', concat_feature=False):
        super().__init__()
        if pos_embedding_method != 'fourier':
            raise NotImplementedError("libadfp implements pos_embedding_method='fourier' only")
        if hidden_size != 32 or n_blocks != 5 or list(skips) != [2] or dim != 3:
            raise NotImplementedError('libadfp implements the hidden=32, 5-block, skips=[2] decoder only')
        self.name = name
        self.color = color
        self.no_grad_feature = False
        self.c_dim = c_dim
        self.grid_len = grid_len
        self.concat_feature = concat_feature
        self.n_blocks = n_blocks
        self.skips = skips
        self.sample_mode = sample_mode
        self.fc_c = nn.ModuleList([nn.Linear(c_dim, hidden_size) for _ in range(n_blocks)])
        emb = 93
        self.embedder = GaussianFourierFeatureTransform(dim, mapping_size=emb, scale=25)
        layers = [DenseLayer(emb, hidden_size, activation='relu')]
        for i in range(n_blocks - 1):
            fan_in = hidden_size + emb if i in skips else hidden_size
            layers.append(DenseLayer(fan_in, hidden_size, activation='relu'))
        self.pts_linears = nn.ModuleList(layers)
        self.output_linear = DenseLayer(hidden_size, 4 if color else 1, activation='linear')

    def forward(self, p, c_grid=None):
        """One decoder on its own (reference decoder.py:177-203; the embedder squeezes the batch dimension, :27): p [1,P,3] -> [P]
        (occupancy decoders) / [P,4] (colour), no bound rule, no TSDF logic.  Inference-only here (see _refuse_autograd); ``self.bound`` must be set like
        src/DF_Prior.py:192-194 does."""
        self._refuse_autograd(p, c_grid, self)
        bound = getattr(self, 'bound', None)
        if bound is None:
            raise RuntimeError(f'{self.name}_decoder.bound is not set (src/DF_Prior.py:192-194 assigns it in load_bound)')
        return _engine().decode_single(self, p.reshape(-1, 3), c_grid, bound)


class mlp_tsdf(_SingleNet, nn.Module):
    """Parameter container of the attention fusion MLP 2->64->128->128->64->2 (decoder.py:206-228)."""

    def __init__(self):
        super().__init__()
        self.no_grad_feature = False
        self.sample_mode = 'bilinear'
        self.pts_linears = nn.ModuleList([DenseLayer(2, 64, activation='relu'),
                                          DenseLayer(64, 128, activation='relu'),
                                          DenseLayer(128, 128, activation='relu'),
                                          DenseLayer(128, 64, activation='relu')])
        self.output_linear = DenseLayer(64, 2, activation='linear')
        self.softmax = nn.Softmax(dim=1)
        self.sigmoid = nn.Sigmoid()

    def forward(self, p, occ, tsdf_volume, tsdf_bnds, **kwargs):
        """The attention fusion on its own (reference decoder.py:240-258): p [1,M,3], occ [M] (or [1,M]) ->
        (fused occupancy [M], attention weight [M]).  Inference-only here."""
        self._refuse_autograd(p, occ, self)
        return _engine().attention_rows(self, p.reshape(-1, 3), occ.reshape(-1), tsdf_volume, tsdf_bnds)


def _flat_params(params):
    """The parameters of one network as one flat float32 tensor in state_dict order.  When they already lie back to back in one
    storage (mapping.flatten_parameters re-homes a trained network that way) this is a view of that storage, not a copy."""
    first = params[0]
    if first.dtype == torch.float32:
        off, ok = first.storage_offset(), True
        for p in params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.storage_offset() != off or \
                    p.untyped_storage().data_ptr() != first.untyped_storage().data_ptr():
                ok = False
                break
            off += p.numel()
        if ok:
            flat = first.detach().new_empty(0)
            flat.set_(first.untyped_storage(), first.storage_offset(), (off - first.storage_offset(),))
            return flat
    return torch.cat([p.detach().reshape(-1).float() for p in params])


_data_ptr = torch.Tensor.data_ptr
_version_of = torch.Tensor._version.__get__


def _version_key(params):
    """(addresses, versions) of a parameter tuple: what a cached conversion of them is valid for."""
    return tuple(map(_data_ptr, params)), tuple(map(_version_of, params))


def _home_in_one_buffer(params):
    """Re-home float32 CUDA parameters in ONE contiguous buffer, state_dict order: every nn.Parameter keeps its identity (optimisers,
    state_dict, deepcopy and pickling see the same objects / names / values) and becomes a view of the buffer, so the flat
    image the pack kernels and the f32 repair path read IS the parameters -- no torch.cat per optimiser step."""
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    off = 0
    for p in params:
        n = p.numel()
        p.data = flat[off:off + n].view(p.shape)
        off += n
    return flat


def home_parameters(module):
    """Puts the float32 CUDA parameters of one sub-network into one buffer (see _SingleNet) unless they are there already.
    Only for a module whose parameter storage is private at this moment (just converted / just copied).  Returns the flat
    view, or None when the parameters are not all float32 on one GPU."""
    params = tuple(module.parameters())
    if not params or not all(p.is_cuda and p.dtype == torch.float32 and p.device == params[0].device for p in params):
        return None
    flat = _flat_params(params)
    if flat.untyped_storage().data_ptr() == params[0].untyped_storage().data_ptr():
        return flat
    return _home_in_one_buffer(params)


_NET_ATTR = {'low': 'low_decoder', 'high': 'high_decoder', 'color': 'color_decoder', 'att': 'mlp'}


class DF(nn.Module):
    """Drop-in for the reference's ``DF`` (decoder.py:262-353).

    ``decoders(p[1,P,3], c_grid=dict, tsdf_volume=, tsdf_bnds=, stage=)`` -> ``(raw[P,4], w[P])``
    exactly as src/utils/Renderer.py:57 and src/utils/Mesher.py:315 call it.  NOTE: like the
    reference's DF.forward this returns the decoder output WITHOUT the out-of-bound rule
    (``ret[~mask,3] = 100`` is applied by the callers, Renderer.py:64 / Mesher.py:322).
    """

    def __init__(self, dim=3, c_dim=32, low_grid_len=0.16, high_grid_len=0.16, color_grid_len=0.16,
                 hidden_size=32, pos_embedding_method='fourier'):
        super().__init__()
        if c_dim != 32:
            raise NotImplementedError('libadfp implements c_dim=32 only (configs/df_prior.yaml:102)')
        self.low_decoder = MLP(name='low', dim=dim, c_dim=c_dim, color=False, skips=[2], n_blocks=5,
                               hidden_size=hidden_size, grid_len=low_grid_len,
                               pos_embedding_method=pos_embedding_method)
        self.high_decoder = MLP(name='high', dim=dim, c_dim=c_dim * 2, color=False, skips=[2], n_blocks=5,
                                hidden_size=hidden_size, grid_len=high_grid_len, concat_feature=True,
                                pos_embedding_method=pos_embedding_method)
        self.color_decoder = MLP(name='color', dim=dim, c_dim=c_dim, color=True, skips=[2], n_blocks=5,
                                 hidden_size=hidden_size, grid_len=color_grid_len,
                                 pos_embedding_method=pos_embedding_method)
        self.mlp = mlp_tsdf()
        self._packed = {}       # name -> (version key, packed tensor)
        self._plists = {}       # name -> tuple of the sub-network's parameters (module walks are slow)
        self._engine = None
        self._status = None     # pinned status word of THIS module (see _lib.new_status_word)
        self._exact_latch = set()   # networks ('low' / 'high' / 'color' / 'att' / 'bwd') switched to the exact f32 kernels
        self._pack_jobs = None      # a list while Engine.scene() collects this call's pack jobs (flush_pack_jobs), else None
        self._foreign = False       # True: the parameters live in ANOTHER process's memory (see __setstate__)
        self._foreign_serial = 0

    def net_params(self, name):
        """The parameters of one sub-network ('low' / 'high' / 'color' / 'att') in state_dict order.  Cached:
        nn.Module.parameters() walks the module tree (0.5 ms per training iteration when called per use);
        conversions that may replace Parameter objects (_apply) drop the cache."""
        hit = self._plists.get(name)
        if hit is None:
            hit = tuple(getattr(self, _NET_ATTR[name]).parameters())
            self._plists[name] = hit
        return hit

    def any_requires_grad(self):
        return any(p.requires_grad for n in _NET_ATTR for p in self.net_params(n))

    def _apply(self, fn, *args, **kwargs):
        self._plists = {}
        return super()._apply(fn, *args, **kwargs)

    # ---- arithmetic mode per network ------------------------------------------------------
    def status_word(self):
        if self._status is None:
            self._status = _lib.new_status_word()
        return self._status

    def absorb_status(self, sync=False, device=None):
        """Reads (and clears) this module's status word.  A network whose f16-split kernels left the f16 range in an earlier call
        -- that call repaired itself on the device, csrc/adfp_fallback.h -- runs on the exact f32 kernels from now on.  The latch
        outlives parameter updates on purpose: one optimiser step does not bring activations of 1e5 back into range, and a
        latch that reset itself every iteration would send every iteration through the slow repair path (with zero gradients).
        ``reset_math_latch()`` / ``load_state_dict`` clear it."""
        v = _lib.read_status(self._status, sync, device)
        if v:
            for name, bit in _lib.STATUS_RANGE_BITS.items():
                if v & bit:
                    self._exact_latch.add(name)
        return v

    def uses_split(self, name):
        """Does network `name` run on the f16-split kernels (ADFP_MATH=f16x3 and not latched to exact)?"""
        from .engine import math_mode
        return math_mode() == 'f16x3' and name not in self._exact_latch

    def reset_math_latch(self):
        self._exact_latch.clear()

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._exact_latch.clear()           # new weights: probe the f16 range again
        return out

    def flat_weights(self, name, key=None):
        """The network's parameters as one flat float32 buffer in state_dict order: a VIEW of the live parameters when they lie
        in one buffer (home_parameters put them there at .to(device) / deepcopy, cached on their addresses), otherwise a copy
        cached on the parameters' versions.  Never moves a parameter: another process may hold the storage through CUDA IPC
        (src/DF_Prior.py:108-110)."""
        module = self.net_params(name)
        if key is None:
            key = _version_key(module)
        hit = self._packed.get(name + '.flat')
        if hit is not None and hit[0] == key[0] and (hit[2] or hit[3] == key[1]):
            return hit[1]
        flat = _flat_params(module)
        is_view = flat.untyped_storage().data_ptr() == module[0].untyped_storage().data_ptr()
        self._packed[name + '.flat'] = (key[0], flat, is_view, key[1])
        return flat

    def net_key(self, name):
        """(addresses, versions) of the network's parameters; hand it to packed_weights / flat_weights to compute it once per call.
        A module RECEIVED from another process (_foreign, __setstate__) never repeats a key: see there."""
        key = _version_key(self.net_params(name))
        if self._foreign:
            self._foreign_serial += 1
            return key[0], (self._foreign_serial,)
        return key

    # ---- weight images for the kernels -------------------------------------------------
    def packed_weights(self, name, fmt='f32', key=None):
        """Packed (MFMA operand order) image of one sub-network, rebuilt only when a parameter
        changed (optimizer step bumps Parameter._version).  fmt 'f32' = exact f32-input MFMA image,
        'h' = f16 hi/lo split image of the forward decoders (adfp_pack_decoder_h), 'ht' = the transposed split image of
        the f16 backward (adfp_pack_decoder_ht / adfp_pack_attention_ht).  key: net_key(name) if the caller has it already.
        A stale image is rebuilt IN PLACE (same address; stream order protects kernels already queued on it)."""
        module = self.net_params(name)
        if key is None:
            key = _version_key(module)
        if fmt in ('h', 'g', 'hg'):
            # ONE buffer, two layouts: 'h' (the training forward, sub-networks on their own) and 'g' (the inference kernels) are
            # kept current separately -- a training iteration re-packs only the H part of a trained network, a frame only the G part
            slot = name + '.split'
            hit = self._packed.get(slot)
            need = [p for p in (('h', 'g') if fmt == 'hg' else (fmt,)) if hit is None or hit[2].get(p) != key]
            if not need:
                return hit[1]
            flat = self.flat_weights(name, key)
            packed = pack_network(name, module, 'hg' if len(need) == 2 else need[0], status=self.status_word(), flat=flat,
                                  out=None if hit is None else hit[1], defer=self._pack_jobs)
            keys = dict(hit[2]) if hit is not None and hit[1] is packed else {}
            for p in need:
                keys[p] = key
            self._packed[slot] = (key, packed, keys)
            return packed
        slot = name if fmt == 'f32' else name + '.' + fmt
        hit = self._packed.get(slot)
        if hit is not None and hit[0] == key:
            return hit[1]
        flat = self.flat_weights(name, key)
        packed = pack_network(name, module, fmt, status=self.status_word(), flat=flat, out=None if hit is None else hit[1],
                              defer=self._pack_jobs if fmt == 'ht' else None)
        self._packed[slot] = (key, packed)
        return packed

    def __deepcopy__(self, memo):
        # src/Tracker.py:144 deep-copies the shared decoders; caches must not be shared
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ('_packed', '_engine', '_plists', '_status', '_pack_jobs'):
                continue
            setattr(new, k, copy.deepcopy(v, memo))
        new._packed = {}
        new._plists = {}
        new._engine = None
        new._status = None
        new._pack_jobs = None
        for m in new.modules():
            m.__dict__.pop('_foreign', None)
        new._foreign, new._foreign_serial = False, 0     # the copy's tensors are this process's own
        for attr in _NET_ATTR.values():            # Parameter.__deepcopy__ clones each tensor on its own; the copy is private here
            home_parameters(getattr(new, attr))
        return new

    def __getstate__(self):
        # torch.multiprocessing spawn pickles the module (src/DF_Prior.py:302-311)
        d = self.__dict__.copy()
        d['_packed'] = {}
        d['_plists'] = {}
        d['_engine'] = None
        d['_status'] = None
        d['_pack_jobs'] = None
        return d

    def __setstate__(self, state):
        """Unpickling = this module arrived from another process (torch.multiprocessing spawn pickles the arguments of
        src/DF_Prior.py:302-311; its CUDA parameters are views of the SENDER's memory through CUDA IPC).  The sender -- the Mapper
        -- trains them in place (src/Mapper.py:364-375), and a write from another process does not move THIS process's tensor
        version counters, which is what every cached conversion of the parameters (packed weight images) is keyed on.  So a
        received module re-packs its images on every call (four pack kernels, ~20 us): what it renders with is what the trainer
        last wrote.  Its deep copies (src/Tracker.py:144 takes one per frame) are private again and cache normally.  The process that
        WRITES the parameters -- the spawned Mapper also receives a pickled copy -- says so with mark_owner()."""
        super().__setstate__(state)              # nn.Module's own: the dict update + its default-attribute fix-ups for older pickles
        self._foreign, self._foreign_serial = True, 0

    def mark_owner(self):
        """This process is the only writer of the parameters (the Mapper process of src/DF_Prior.py:302-311, which trains the shared
        decoders in place through its own optimiser): its version counters see every update, so the packed images and the flat
        views cache on them again instead of being rebuilt per call.  mapping.MapperIteration calls it for the decoders it is
        bound to; a caller that drives `torch.optim` itself on a received module calls it once.  Returns self."""
        self._foreign, self._foreign_serial = False, 0
        for m in self.modules():
            m.__dict__.pop('_foreign', None)
            m.__dict__.pop('_single', None)
        self._packed = {}
        return self

    def forward(self, p, c_grid, tsdf_volume, tsdf_bnds, stage='low', **kwargs):
        from .engine import Engine
        if self._engine is None:
            self._engine = Engine()
        bound = getattr(self, 'bound', None)
        if bound is None:
            raise RuntimeError('DF.bound is not set (src/DF_Prior.py:191 assigns it in load_bound)')
        pts = p.reshape(-1, 3)
        raw, w = self._engine.eval_points(self, pts, c_grid, tsdf_volume, tsdf_bnds, bound, stage,
                                          apply_bound_rule=False)
        return raw, w
