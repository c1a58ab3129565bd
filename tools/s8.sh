cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/s8_pytest.txt 2>&1; grep -n "passed\|failed" $O/s8_pytest.txt
python tools/shard_step.py 2>/dev/null > $O/s8_shard_pipelined.txt; tail -5 $O/s8_shard_pipelined.txt
ADFP_HOST_TIMING=1 python tools/host_breakdown.py --rays 1000 5000 > $O/s8_host_breakdown.txt 2>&1; grep -v "calls      0.0 us" $O/s8_host_breakdown.txt | head -70
python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 300 --graph 2>/dev/null | tail -1
python bench.py --steps 20 --warmup 5 --cpu-rays 0 --no-extra --no-stage-timing 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('headline %.3f ms, %.2f M rays/s; k8 bound incl gather %.3f; k8 shard %.4f ms' % (r['ms_per_step'], r['value']/1e6, r['config']['k8_speedup_bound_incl_gather'], r['config']['shard_model']['k8']['ms_slowest_shard']))"
