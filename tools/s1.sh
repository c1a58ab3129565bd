# round-6 session 1: baseline of the 1/8 shard (pipelined time + kernel timeline) and the weight-gate histogram
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
python tools/shard_step.py > $O/s1_shard_pipelined.txt 2>&1
for kr in "8 3" "1 0"; do
  set -- $kr
  rm -rf $O/prof_shard
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_shard -- python3 tools/shard_step.py --trace $1 $2 > $O/s1_shard${1}_trace.log 2>&1
  f=$(find $O/prof_shard -name '*kernel_trace.csv' | head -1)
  python tools/trace_timeline.py $f k_get_rays > $O/s1_shard${1}_timeline.txt 2>&1
  python profiles/summarize.py $O/prof_shard $O/s1_kernel_stats_shard${1}.csv > /dev/null 2>&1
done
rm -rf $O/prof_shard
python tools/weight_histogram.py > $O/s1_weight_histogram.txt 2>&1
tail -3 $O/s1_shard_pipelined.txt; tail -12 $O/s1_weight_histogram.txt
