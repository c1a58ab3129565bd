# round-6 session 2: GPU suite on the frame-job build, the 1/8 shard again, a short bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/s2_pytest.txt 2>&1; tail -5 $O/s2_pytest.txt
python tools/shard_step.py > $O/s2_shard_pipelined.txt 2>&1; tail -5 $O/s2_shard_pipelined.txt
rm -rf $O/prof_shard
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_shard -- python3 tools/shard_step.py --trace 8 3 > $O/s2_shard8_trace.log 2>&1
f=$(find $O/prof_shard -name '*kernel_trace.csv' | head -1)
python tools/trace_timeline.py $f k_relayout_cm_to_cl > $O/s2_shard8_timeline.txt 2>&1; cat $O/s2_shard8_timeline.txt
rm -rf $O/prof_shard
python bench.py --steps 20 --warmup 5 --cpu-rays 4000 --no-extra > $O/s2_bench.json 2> $O/s2_bench.err; tail -c 1500 $O/s2_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/s2_bench.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'])
print(json.dumps(d['config'], indent=1)[:3000])
print(json.dumps(d['roofline'].get('limiter'), indent=1)[:1500])
PY
