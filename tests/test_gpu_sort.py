"""GPU: adfp_sort_pairs (the hand-written stable LSD radix sort that orders the sample points by grid cell for the backward's
scatter, csrc/adfp_sort.h) against torch.sort(stable=True): sizes around the 1 024-key tile and its multiples, one to four passes, all-equal
keys, a handful of hot keys (what a camera frustum produces)."""
import pytest
import torch

from attentive_dfprior_amd import _lib
from attentive_dfprior_amd._lib import lib, ptr, check

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def sort_pairs(key, val, bits):
    L = lib()
    n = key.numel()
    k, v = key.clone(), val.clone()
    kt, vt = torch.empty_like(k), torch.empty_like(v)
    ws = torch.empty(max(int(L.adfp_sort_workspace_bytes(n)), 4), dtype=torch.uint8, device=DEV)
    with torch.cuda.device(DEV):
        check(L.adfp_sort_pairs(ptr(k), ptr(v), ptr(kt), ptr(vt), n, bits, ptr(ws), ws.numel(), _lib.current_stream(DEV)), 'adfp_sort_pairs')
    return k, v


@pytest.mark.parametrize('n', [1, 63, 64, 1023, 1024, 1025, 2047, 2048, 2049, 5000, 320000, 1000003])
@pytest.mark.parametrize('bits', [7, 8, 15, 17, 21, 24, 30])
def test_sort_pairs_is_a_stable_sort(n, bits):
    g = torch.Generator(device='cpu').manual_seed(n * 31 + bits)
    key = torch.randint(0, 2 ** bits, (n,), generator=g, dtype=torch.int64).to(torch.int32).to(DEV)
    val = torch.arange(n, dtype=torch.int32, device=DEV)
    k, v = sort_pairs(key, val, bits)
    rk, order = torch.sort(key.long(), stable=True)
    assert torch.equal(k.long(), rk)
    assert torch.equal(v.long(), order)              # stable: equal keys keep their input order


@pytest.mark.parametrize('kind', ['equal', 'hot'])
def test_sort_pairs_with_crowded_keys(kind):
    n = 200000
    g = torch.Generator(device='cpu').manual_seed(3)
    if kind == 'equal':
        key = torch.full((n,), 12345, dtype=torch.int32)
    else:                                            # 90 % of the keys from 5 values
        hot = torch.tensor([7, 70000, 70001, 1 << 20, (1 << 21) + 3])
        key = torch.where(torch.rand(n, generator=g) < 0.9, hot[torch.randint(0, 5, (n,), generator=g)],
                          torch.randint(0, 1 << 22, (n,), generator=g)).to(torch.int32)
    key = key.to(DEV)
    val = torch.arange(n, dtype=torch.int32, device=DEV)
    k, v = sort_pairs(key, val, 22)
    rk, order = torch.sort(key.long(), stable=True)
    assert torch.equal(k.long(), rk) and torch.equal(v.long(), order)
