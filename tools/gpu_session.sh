cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/s25_pytest.log 2>&1
echo "pytest rc=$?" >> $O/s25_pytest.log; grep -n "passed\|failed" $O/s25_pytest.log | tail -2; tail -30 $O/s25_pytest.log | grep -v "^$" | tail -12
for rep in 1 2; do
  python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 200 --graph 2>&1 | tail -1 | sed "s/^/graph 5000x64: /"
  python tools/profile_iteration.py --rays 1000 --samples 32 --masked --iters 200 --graph 2>&1 | tail -1 | sed "s/^/graph 1000x48: /"
done > $O/s25_iter.txt 2>&1; cat $O/s25_iter.txt
rm -rf /tmp/pr; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 30 > /dev/null 2>&1
python profiles/summarize.py /tmp/pr $O/s25_kernel_stats_train.csv | grep "k_decode_high_g\|k_decode_h<64\|k_attention_h"
