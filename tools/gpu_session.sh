cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_grad.py tests/test_gpu_mapper_iteration.py tests/test_gpu_config1.py tests/test_gpu_tracker_iteration.py tests/test_gpu_mapping.py tests/test_gpu_graph.py tests/test_gpu_config3.py tests/test_gpu_dist2.py tests/test_gpu_tsdf_blocks.py -m gpu -q -p no:cacheprovider -x > $O/s16_pytest.log 2>&1
echo "pytest rc=$?" >> $O/s16_pytest.log; tail -5 $O/s16_pytest.log
for rep in 1 2; do
  python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 200 --graph 2>&1 | tail -1 | sed "s/^/graph 5000x64: /"
  python tools/profile_iteration.py --rays 1000 --samples 32 --masked --iters 200 --graph 2>&1 | tail -1 | sed "s/^/graph 1000x48: /"
done > $O/s16_iter.txt 2>&1; cat $O/s16_iter.txt
rm -rf /tmp/pr; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 30 > /dev/null 2>&1
python profiles/summarize.py /tmp/pr $O/s16_kernel_stats_train.csv | head -32 | cut -c1-120
