"""GPU: where the decoder parameters live, and that nobody moves them behind a sharer's back.

The reference moves the decoders to the GPU once (src/DF_Prior.py:50-51), calls share_memory() (:108-110) and hands the module to
its Mapper and Tracker PROCESSES (:302-311: CUDA IPC).  The Mapper then trains the colour decoder and the attention MLP in place
(src/Mapper.py:364-375) and the Tracker deep-copies the shared module every frame (src/Tracker.py:144).  A renderer that re-homes
`p.data` inside a render call cuts that link: the optimiser writes to the new memory, the other process keeps reading the old one.
So the flat buffer the kernels want is set up where the storage is replaced anyway -- `.to(device)` and `deepcopy` -- and a
render call never changes `data_ptr()` of any parameter."""
import copy
import os
import sys

import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import synthetic
from oracle import adfp_oracle as O
from conftest import make_cfg, to_dev, assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _setup(n=64):
    sc = synthetic.mini_scene()
    sd = O.random_state_dict(seed=3)
    ro, rd, gd, _ = synthetic.make_ray_batch(sc, n, seed=5)
    dec = A.DF()
    dec.load_state_dict(sd)
    dec.bound = sc.bound
    rend = A.Renderer(make_cfg(32, 16), None, sc)
    return sc, sd, (ro, rd, gd), dec, rend


def _render(rend, dec, sc, rays, stage='color'):
    ro, rd, gd = rays
    with torch.no_grad():
        return rend.render_batch_ray(to_dev(sc.c, DEV), dec, rd.to(DEV), ro.to(DEV), DEV, sc.tsdf_volume.to(DEV),
                                     sc.tsdf_bnds.to(DEV), stage, gt_depth=gd.to(DEV))


def _one_buffer(module):
    ps = list(module.parameters())
    base = ps[0].untyped_storage().data_ptr()
    off = ps[0].storage_offset()
    for p in ps:
        if p.untyped_storage().data_ptr() != base or p.storage_offset() != off or not p.is_contiguous():
            return False
        off += p.numel()
    return True


def test_to_device_homes_every_network_and_render_moves_nothing():
    sc, sd, rays, dec, rend = _setup()
    dec = dec.to(DEV)
    for attr in ('low_decoder', 'high_decoder', 'color_decoder', 'mlp'):
        assert _one_buffer(getattr(dec, attr)), f'{attr}: .to(device) leaves the parameters in one buffer, state_dict order'
    before = {n: (p.data_ptr(), p.untyped_storage().data_ptr()) for n, p in dec.named_parameters()}
    dec.share_memory()                                                    # src/DF_Prior.py:108-110: must not move anything either
    out = _render(rend, dec, sc, rays)
    torch.cuda.synchronize()
    after = {n: (p.data_ptr(), p.untyped_storage().data_ptr()) for n, p in dec.named_parameters()}
    assert before == after, 'a render call must not re-home a parameter'
    ref = O.render_batch_ray(sd, sc.c, rays[1], rays[0], sc.tsdf_volume, sc.tsdf_bnds, sc.bound, 'color', rays[2], 32, 16)
    for got, want, name in zip(out, ref, ('depth', 'uncertainty', 'color', 'weight')):
        assert_close(got, want, 1e-4, name)
    # state_dict / load_state_dict / deepcopy keep working on the views
    sd2 = {k: v.clone() for k, v in dec.state_dict().items()}
    assert set(sd2) == set(sd)
    cp = copy.deepcopy(dec)                                                # src/Tracker.py:144
    for attr in ('low_decoder', 'high_decoder', 'color_decoder', 'mlp'):
        assert _one_buffer(getattr(cp, attr))
    assert all(a.data_ptr() != b.data_ptr() for a, b in zip(cp.parameters(), dec.parameters())), 'the copy owns its memory'
    out2 = _render(rend, cp, sc, rays)
    for a, b in zip(out, out2):
        assert torch.equal(a, b)


def test_scattered_parameters_stay_where_they_are_and_updates_are_seen():
    """Parameters somebody else placed (p.data assigned tensor by tensor: another framework's loader, a sharer's views) are served
    from a copy cached on their versions: pointers unchanged, in-place updates picked up by the next call."""
    sc, sd, rays, dec, rend = _setup()
    dec = dec.to(DEV)
    for p in dec.parameters():
        p.data = p.data.clone()                                           # every tensor its own allocation
    assert not _one_buffer(dec.color_decoder)
    before = {n: p.data_ptr() for n, p in dec.named_parameters()}
    out = _render(rend, dec, sc, rays)
    assert before == {n: p.data_ptr() for n, p in dec.named_parameters()}
    ref = O.render_batch_ray(sd, sc.c, rays[1], rays[0], sc.tsdf_volume, sc.tsdf_bnds, sc.bound, 'color', rays[2], 32, 16)
    assert_close(out[2], ref[2], 1e-4, 'colour, scattered parameters')
    with torch.no_grad():
        dec.color_decoder.output_linear.bias.add_(0.25)                  # an optimiser step in place
    sd_new = {k: v.clone() for k, v in sd.items()}
    sd_new['color_decoder.output_linear.bias'] += 0.25
    out = _render(rend, dec, sc, rays)
    ref = O.render_batch_ray(sd_new, sc.c, rays[1], rays[0], sc.tsdf_volume, sc.tsdf_bnds, sc.bound, 'color', rays[2], 32, 16)
    assert_close(out[2], ref[2], 1e-4, 'colour after the in-place update')


def _tracker_process(dec, pipe, rays_np, steps):
    """The other process of src/DF_Prior.py:302-311: renders with the SHARED module (and with a deep copy of it, Tracker.py:144)
    each time the parent says so and sends the colour image back (as numpy: only the module travels as tensors, through CUDA IPC)."""
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        rays = tuple(torch.from_numpy(r) for r in rays_np)
        sc = synthetic.mini_scene()
        rend = A.Renderer(make_cfg(32, 16), None, sc)
        for _ in range(steps):
            pipe.recv()
            shared = _render(rend, dec, sc, rays)[2].cpu().numpy()
            copied = _render(rend, copy.deepcopy(dec), sc, rays)[2].cpu().numpy()
            pipe.send((shared, copied))
    except Exception as e:                                               # the parent turns it into a failure (or a skip for IPC)
        pipe.send(RuntimeError(f'{type(e).__name__}: {e}'))


def test_two_processes_share_the_parameters_through_ipc():
    """Process A (the Mapper) renders -- which used to re-home its parameters -- then steps a parameter in place; process B (the
    Tracker) must see the step, through the shared module and through its per-frame deep copy."""
    import torch.multiprocessing as mp
    sc, sd, rays, dec, rend = _setup(n=32)
    dec = dec.to(DEV)
    dec.share_memory()
    ctx = mp.get_context('spawn')
    here, there = ctx.Pipe()
    steps = 2
    # The module also carries a CPU tensor (`bound`, as src/DF_Prior.py:191 assigns it).  CPU tensors of a spawn argument travel as
    # file descriptors served by a socket in the parent's multiprocessing temp directory -- which a FORKED child of an earlier test
    # removes on its exit (its inherited finalizers run).  The file-system strategy needs no server; set for this spawn only.
    strategy = mp.get_sharing_strategy()
    mp.set_sharing_strategy('file_system')
    try:
        proc = ctx.Process(target=_tracker_process, args=(dec, there, tuple(r.numpy() for r in rays), steps))
        proc.start()
    except RuntimeError as e:                                            # no CUDA IPC for this user / driver
        pytest.skip(f'CUDA IPC not available here: {e}')
    finally:
        mp.set_sharing_strategy(strategy)
    try:
        sd_now = {k: v.clone() for k, v in sd.items()}
        for step in range(steps):
            mine = _render(rend, dec, sc, rays)[2].cpu()               # the parent's own render call first
            here.send('go')
            if not here.poll(300):
                pytest.fail('the child process did not answer')
            got = here.recv()
            if isinstance(got, Exception):
                if 'ipc' in str(got).lower():
                    pytest.skip(f'CUDA IPC not available here: {got}')
                raise got
            ref = O.render_batch_ray(sd_now, sc.c, rays[1], rays[0], sc.tsdf_volume, sc.tsdf_bnds, sc.bound, 'color', rays[2], 32, 16)[2]
            assert_close(mine, ref, 1e-4, f'step {step}: parent')
            assert_close(got[0], ref, 1e-4, f'step {step}: child, shared module')
            assert_close(got[1], ref, 1e-4, f'step {step}: child, deep copy')
            with torch.no_grad():                                        # the Mapper's optimiser step, in place
                dec.color_decoder.output_linear.bias.add_(0.5)
                dec.mlp.output_linear.bias.add_(torch.tensor([0.3, -0.3], device=DEV))
            torch.cuda.synchronize()
            sd_now['color_decoder.output_linear.bias'] = sd_now['color_decoder.output_linear.bias'] + 0.5
            sd_now['mlp.output_linear.bias'] = sd_now['mlp.output_linear.bias'] + torch.tensor([0.3, -0.3])
    finally:
        proc.join(60)
        if proc.is_alive():
            proc.kill()
