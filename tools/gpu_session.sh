cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_grad.py tests/test_gpu_mapper_iteration.py tests/test_gpu_config3.py tests/test_gpu_dist2.py tests/test_gpu_scale.py -m gpu -q -p no:cacheprovider -x > $O/s8_pytest.log 2>&1
echo "pytest rc=$?" >> $O/s8_pytest.log; tail -4 $O/s8_pytest.log
ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_roles_span.so python tools/roles_span.py 2>&1 | tail -3 > $O/s8_roles_span.txt
cat $O/s8_roles_span.txt
for rep in 1 2; do
  python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 200 --graph 2>&1 | tail -1 | sed "s/^/role-split  graph 5000x64: /"
  python tools/profile_iteration.py --rays 1000 --samples 32 --masked --iters 200 --graph 2>&1 | tail -1 | sed "s/^/role-split  graph 1000x48: /"
done > $O/s8_iter.txt 2>&1; cat $O/s8_iter.txt
rm -rf /tmp/pr; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 30 > /dev/null 2>&1
python profiles/summarize.py /tmp/pr $O/s8_kernel_stats_train.csv | head -60
