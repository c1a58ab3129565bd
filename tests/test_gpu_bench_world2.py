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
    # the headline at N > 1 is the RAY-SHARDED frame (strong scaling, the all-gather inside the timed region): one 640x480 frame per
    # step for the whole job, whatever N is -- not one frame per rank
    assert r['n_gpus'] == 2 and r['scaling'] == 'strong' and r['value'] > 0
    assert r['config']['rays_per_step'] == 640 * 480 and r['config']['rays_per_step_per_gpu'] == 640 * 480 // 2
    assert abs(r['value'] - 640 * 480 / (r['ms_per_step'] * 1e-3)) <= 1e-6 * r['value']
    assert 'ray-sharded over the 2 GPUs' in r['config']['workload']
    # ... and the old headline (one whole frame per rank, no collective) is kept beside it under its own name
    assert r['weak']['scaling'] == 'weak' and r['weak']['n_gpus'] == 2 and r['weak']['value'] > 0
    assert abs(r['weak']['value'] - 2 * 640 * 480 / (r['weak']['ms_per_step'] * 1e-3)) <= 1e-6 * r['weak']['value']
    assert 'dist_legs_error' not in r, r.get('dist_legs_error')
    assert r['rccl']['world_size'] == 2 and r['rccl']['allreduce_of_ones'] == 2.0
    assert r['strong']['n_gpus'] == 2 and r['strong']['value'] > 0
    t = r['train_allreduce']
    assert t['n_gpus'] == 2 and t['rays_per_rank'] == t['rays_per_iteration'] // 2
    for leg in ('dense', 'frustum_masked', 'fused_dense', 'fused_frustum_masked'):
        assert t[leg]['ms_per_iteration'] > 0
    assert t['fused_frustum_masked']['bucket_bytes'] < t['fused_dense']['bucket_bytes']


def test_bench_one_gpu_line_carries_the_shard_model():
    """N = 1: `config.shard_model` times a 1/2, 1/4 and 1/8 shard of the headline frame (per-step fixed costs included) and states the
    efficiency bound they imply -- the north star's >= 6x at 8 GPUs bounded from one GPU."""
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--cpu-rays', '0', '--no-extra', '--no-stage-timing']
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r['n_gpus'] == 1 and r['scaling'] == 'strong' and 'weak' not in r
    m = r['config']['shard_model']
    assert 'error' not in m, m
    assert m['k1']['ms'] > 0
    # the numerator is the HEADLINE's ms_per_step (what a SCALE run divides by), the shards are timed pipelined like it, and the
    # all-gather is in the bound as a ring model (5.49 / (0.937 + 0.084) = 5.4x was printed as 6.24x)
    assert m['headline_ms_per_step'] == r['ms_per_step']
    assert m['allgather_bytes'] == 640 * 480 * 28
    for k in (2, 4, 8):
        e = m[f'k{k}']
        assert 0 < e['ms_fastest_shard'] <= e['ms_slowest_shard'] < m['k1']['ms'] * 1.05
        g = ((k - 1) / k * m['allgather_bytes'] / 153e9 + (k - 1) * 5e-6) * 1e3
        assert abs(e['ms_allgather_model'] - g) < 1e-9
        assert abs(e['speedup_bound'] - r['ms_per_step'] / e['ms_slowest_shard']) < 1e-9
        assert abs(e['speedup_bound_incl_gather'] - r['ms_per_step'] / (e['ms_slowest_shard'] + g)) < 1e-9
        assert 0 < e['speedup_bound_incl_gather'] < e['speedup_bound'] <= 1.3 * k
    assert m['fixed_cost_ms'] > 0
    # a scalar directly under `config` (the driver's record drops dict-valued keys), and a verdict derived from THAT number
    b8 = r['config']['k8_speedup_bound_incl_gather']
    assert b8 == m['k8']['speedup_bound_incl_gather']
    assert m['verdict'].startswith('reachable' if b8 >= 6.0 else 'NOT reachable') and f'{b8:.2f}x' in m['verdict']
