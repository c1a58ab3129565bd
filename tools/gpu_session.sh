cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pack_images.py tests/test_gpu_split_images.py tests/test_gpu_grad.py tests/test_gpu_mapper_iteration.py tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_gpu_tracker_iteration.py -m gpu -q -p no:cacheprovider -x > $O/s10_pytest.log 2>&1
echo "pytest rc=$?" >> $O/s10_pytest.log; tail -5 $O/s10_pytest.log
ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_roles_span.so python tools/roles_span.py 2>&1 | tail -3 > $O/s10_roles_span.txt
cat $O/s10_roles_span.txt
for rep in 1 2; do
  python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 200 --graph 2>&1 | tail -1 | sed "s/^/in-tree      graph 5000x64: /"
  ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_roles_prio.so python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 200 --graph 2>&1 | tail -1 | sed "s/^/young-prio   graph 5000x64: /"
  python tools/profile_iteration.py --rays 1000 --samples 32 --masked --iters 200 --graph 2>&1 | tail -1 | sed "s/^/in-tree      graph 1000x48: /"
done > $O/s10_iter.txt 2>&1; cat $O/s10_iter.txt
rm -rf /tmp/pr; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 30 > /dev/null 2>&1
python profiles/summarize.py /tmp/pr $O/s10_kernel_stats_train.csv | head -50
