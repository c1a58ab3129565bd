cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/s21_pytest.log 2>&1
echo "pytest rc=$?" >> $O/s21_pytest.log; tail -4 $O/s21_pytest.log
for rep in 1 2; do
  python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 200 --graph 2>&1 | tail -1 | sed "s/^/graph 5000x64: /"
  python tools/profile_iteration.py --rays 1000 --samples 32 --masked --iters 200 --graph 2>&1 | tail -1 | sed "s/^/graph 1000x48: /"
done > $O/s21_iter.txt 2>&1; cat $O/s21_iter.txt
rm -rf /tmp/pr; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 30 > /dev/null 2>&1
python profiles/summarize.py /tmp/pr $O/s21_kernel_stats_train.csv | grep "k_forward_head\|k_adam_step\|k_backward_head\|k_pack_multi\|k_zero_multi\|k_adam"
