"""GPU: `python bench.py --gpus 2` end to end on a box with ONE MI355X.  RCCL refuses two ranks on one device, so the test
hooks of bench.py put both ranks on cuda:0 and swap the backend for gloo; everything else is the code the driver runs at
N = 2 / 4 / 8: the parent launches the ranks before touching a GPU, every rank renders its frame, barrier + max over ranks,
the strong-scaling frame (contiguous ray slices, packed all-gather) and the sharded training iterations with the gradient
bucket all-reduce.  Asserts that the ONE JSON line reports the world it ran in and that the legs produced numbers."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_two_ranks_and_reports_them():
    env = dict(os.environ, ADFP_BENCH_TEST_SAME_DEVICE='1', ADFP_BENCH_TEST_BACKEND='gloo')
    env.pop('WORLD_SIZE', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--cpu-rays', '0']
    # The run takes ~10 s.  Once in ~25 leases the two-process gloo rendezvous / first collective on a fresh box did not complete
    # (observed once in round 4, not reproducible in 12 consecutive runs on another box): a run that exceeds 5 minutes is started
    # again ONCE; a wrong result or a non-zero exit is never retried.
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    except subprocess.TimeoutExpired as e:
        print('bench.py --gpus 2 timed out once; stderr tail:', (e.stderr or b'').decode(errors='replace')[-1500:])
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and r['scaling'] == 'weak' and r['value'] > 0
    assert 'dist_legs_error' not in r, r.get('dist_legs_error')
    assert r['rccl']['world_size'] == 2 and r['rccl']['allreduce_of_ones'] == 2.0
    assert r['strong']['n_gpus'] == 2 and r['strong']['value'] > 0
    t = r['train_allreduce']
    assert t['n_gpus'] == 2 and t['rays_per_rank'] == t['rays_per_iteration'] // 2
    for leg in ('dense', 'frustum_masked', 'fused_dense', 'fused_frustum_masked'):
        assert t[leg]['ms_per_iteration'] > 0
    assert t['fused_frustum_masked']['bucket_bytes'] < t['fused_dense']['bucket_bytes']
