"""GPU: the transposed f16-split weight images of the backward (adfp_pack_decoder_ht / adfp_pack_attention_ht) decoded on the
host and compared with the parameters they were packed from.  The layout is restated here independently of the kernels:
(weights of ~0.1 have lo halves in the f16 subnormal range: hi + lo reproduces them to ~3e-8 absolute)
a chain block is 2 k-steps x [hi | lo] x [lane half h] x [32 rows] x [8 halves]; k-step ks, half h, element j carries OUT
unit kmap(8 ks + j, h) of the block's out-block, the row is the IN unit -- i.e. image(block)[row][k] = W[out][in]."""
import numpy as np
import pytest
import torch

import attentive_dfprior_amd as A
from attentive_dfprior_amd import _lib
from oracle import adfp_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def kmap(s, h):
    return (s & 3) + 8 * (s >> 2) + 4 * h


def decode_block(words, base):
    """1024 words at `base` -> float64 [32 rows][32 out units] = hi + lo of every weight of the block."""
    w = words[base:base + 1024].view(np.uint16).astype(np.uint16).view(np.float16).astype(np.float64)      # 2048 halves
    out = np.zeros((32, 32))
    for ks in range(2):
        for h in range(2):
            for j in range(8):
                o = kmap(8 * ks + j, h)
                for part in range(2):
                    idx = (((ks * 2 + part) * 2 + h) * 32 + np.arange(32)) * 8 + j
                    out[:, o] += w[idx]
    return out


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize('name', ['low', 'high', 'color'])
def test_decoder_t_image_holds_the_transposed_weights(name):
    dec = A.DF()
    dec.load_state_dict(O.random_state_dict(seed=5))
    dec = dec.to(DEV)
    words = dec.packed_weights(name, 'ht').cpu().numpy().view(np.uint32)
    net = {'low': dec.low_decoder, 'high': dec.high_decoder, 'color': dec.color_decoder}[name]
    sd = {k: v.detach().cpu().double().numpy() for k, v in net.state_dict().items()}
    assert np.array_equal(words[:384].view(np.float32).reshape(96, 4)[:93, :3], sd['embedder._B'].T.astype(np.float32))
    base = 384
    for i in range(5):
        wc = sd[f'fc_c.{i}.weight']                       # [32 out][c_dim in]; the own grid's channels are the first 32
        assert rel(decode_block(words, base), wc[:, :32].T) < 1e-6, f'fc_c.{i}'
        base += 1024
        wp = sd[f'pts_linears.{i}.weight']                # [32 out][in_dim]
        nb = 3 if i == 0 else (4 if i == 3 else 1)
        for ib in range(nb):
            if i == 0 or (i == 3 and ib < 3):
                cols = np.arange(32 * ib, 32 * ib + 32)
                ref = np.where(cols[:, None] < 93, wp[:, np.minimum(cols, 92)].T, 0.0)       # features 93..95 are padding
            elif i == 3:
                ref = wp[:, 93:125].T
            else:
                ref = wp.T
            assert rel(decode_block(words, base), ref) < 1e-6, f'pts_linears.{i} in-block {ib}'
            base += 1024
    assert base + 2 * sd['output_linear.weight'].shape[0] * 16 == words.size


def test_attention_t_image_holds_the_transposed_weights():
    dec = A.DF()
    dec.load_state_dict(O.random_state_dict(seed=5))
    dec = dec.to(DEV)
    words = dec.packed_weights('att', 'ht').cpu().numpy().view(np.uint32)
    sd = {k: v.detach().cpu().double().numpy() for k, v in dec.mlp.state_dict().items()}
    base = 256
    for key, nib, nob in (('pts_linears.3.weight', 4, 2), ('pts_linears.2.weight', 4, 4), ('pts_linears.1.weight', 2, 4)):
        w = sd[key]                                        # [out][in]
        for ib in range(nib):
            for ob in range(nob):
                assert rel(decode_block(words, base), w[32 * ob:32 * ob + 32, 32 * ib:32 * ib + 32].T) < 1e-6, (key, ib, ob)
                base += 1024
    assert base + 128 == words.size == _lib.lib().adfp_attention_packed_ht_words()


def test_pack_images_in_one_launch_equals_the_single_image_entries():
    """adfp_pack_images (several images of several networks in ONE launch: what Engine.scene() issues per call) against
    adfp_pack_split_image / adfp_pack_decoder_ht / adfp_pack_attention_ht one by one: every image bit for bit, and the same
    range report for a weight the f16 split cannot hold."""
    import ctypes as C
    from attentive_dfprior_amd.decoder import pack_network, flush_pack_jobs
    sd = O.random_state_dict(seed=9)
    sd['mlp.pts_linears.2.weight'][3, 5] = 9.0e4                          # beyond the f16 range: flagged by both paths
    dec = A.DF()
    dec.load_state_dict(sd)
    dec = dec.to(DEV)
    cases = [('low', 'g'), ('high', 'hg'), ('color', 'g'), ('color', 'ht'), ('att', 'h'), ('att', 'ht')]      # 7 jobs (one 'hg' = two)
    st_one, st_all = _lib.new_status_word(), _lib.new_status_word()
    single = {(n, f): pack_network(n, dec.net_params(n), f, status=st_one, flat=dec.flat_weights(n)) for n, f in cases}
    jobs, batched = [], {}
    for n, f in cases:
        out = torch.full_like(single[(n, f)], 0x5A5A5A5A)                # only the requested part(s) may be written
        batched[(n, f)] = pack_network(n, dec.net_params(n), f, status=st_all, flat=dec.flat_weights(n), out=out, defer=jobs)
    assert len(jobs) == 7
    flush_pack_jobs(jobs, st_all, DEV)
    torch.cuda.synchronize()
    assert jobs == []
    for (n, f), ref in single.items():
        got = batched[(n, f)]
        if f in ('hg', 'ht'):
            assert torch.equal(got, ref), (n, f)
        else:                                                            # 'h' / 'g': the other part of the buffer is not this job's
            nh = int(_lib.lib().adfp_decoder_packed_h_words(_lib.DEC_KIND[n])) if n != 'att' else int(_lib.lib().adfp_attention_packed_h_words())
            diff = torch.nonzero(got != ref).reshape(-1)
            ref_untouched = torch.nonzero(ref != single[(n, f)]).numel() == 0
            assert ref_untouched
            # the single-image entry left the unrequested part uninitialised (torch.empty): compare where the batched pack wrote
            wrote = torch.nonzero(got != 0x5A5A5A5A).reshape(-1)
            assert wrote.numel() > 1000 and torch.equal(got[wrote], ref[wrote]), (n, f)
            assert nh > 0
    assert int(st_one[0]) == int(st_all[0]) == _lib.STATUS_RANGE_BITS['att']
    # and the engine's calls go through it: a training forward of a fresh module packs everything it and its backward need at once
    from attentive_dfprior_amd import synthetic
    from conftest import make_cfg
    sc = synthetic.mini_scene()
    dec2 = A.DF(); dec2.load_state_dict(O.random_state_dict(seed=3)); dec2.bound = sc.bound; dec2 = dec2.to(DEV)
    rend = A.Renderer(make_cfg(32, 16), None, sc)
    ro, rd, gd, gc = [t.to(DEV) for t in synthetic.make_ray_batch(sc, 64, seed=5)]
    c = {k: v.to(DEV).clone().requires_grad_(True) for k, v in sc.c.items()}
    d, u, col, w = rend.render_batch_ray(c, dec2, rd, ro, DEV, sc.tsdf_volume.to(DEV), sc.tsdf_bnds.to(DEV), 'color', gt_depth=gd)
    keys = set(dec2._packed)
    assert {'color.ht', 'att.ht', 'low.ht', 'high.ht'} <= keys, keys         # the backward's images were packed with the forward's
    (d.sum() + col.sum()).backward()
    assert all(torch.isfinite(p.grad).all() for p in dec2.parameters() if p.grad is not None)
