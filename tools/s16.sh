cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/s16_pytest.txt 2>&1; grep -n "passed\|failed" $O/s16_pytest.txt; grep -n "Error\|assert " $O/s16_pytest.txt | head -20
python tools/shard_step.py 2>/dev/null | tail -5
rm -rf $O/prof_shard
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_shard -- python3 tools/shard_step.py --trace 8 3 > $O/s16_trace.log 2>&1
f=$(find $O/prof_shard -name '*kernel_trace.csv' | head -1)
python tools/trace_timeline.py $f k_forward_head; rm -rf $O/prof_shard
for lib in "" $PWD/tools/ab_r06a/libadfp.so ""; do
ADFP_LIB_PATH=$lib python bench.py --steps 20 --warmup 5 --cpu-rays 0 --no-extra --no-stage-timing 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('lib %-40s headline %.3f ms, %.2f M rays/s; k8 bound incl gather %.3f; k8 shard %.4f ms' % ('$lib' or 'in-tree (sampler + TSDF fused)', r['ms_per_step'], r['value']/1e6, r['config']['k8_speedup_bound_incl_gather'], r['config']['shard_model']['k8']['ms_slowest_shard']))"
done
for rep in 1 2; do
python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 300 --graph 2>/dev/null | tail -1 | sed "s/^/fused sampler, 5000 x 64: /"
ADFP_LIB_PATH=$PWD/tools/ab_r06a/libadfp.so python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 300 --graph 2>/dev/null | tail -1 | sed "s/^/separate k_sample, 5000 x 64: /"
done
