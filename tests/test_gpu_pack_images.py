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
