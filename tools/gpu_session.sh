cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_grad.py tests/test_gpu_mapper_iteration.py tests/test_gpu_config1.py tests/test_gpu_mapping.py tests/test_gpu_dist2.py tests/test_gpu_config3.py -m gpu -q -p no:cacheprovider -x > $O/s24_pytest.log 2>&1
echo "pytest rc=$?" >> $O/s24_pytest.log; tail -3 $O/s24_pytest.log
kstat() {
  rm -rf /tmp/pr; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 40 > /dev/null 2>&1
  python - "$1" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/pr/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_scatter_sorted' in r['Name']:
        print(sys.argv[1], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs']) / 1e3, 1), 'min', round(float(r['MinNs']) / 1e3, 1))
PY
}
{ kstat "in-tree (PPW 64, NW 4):"
for v in sc_64_2 sc_64_3 sc_64_6; do ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_$v.so kstat "$v:"; done
kstat "in-tree again:"; } > $O/s24.txt 2>&1; cat $O/s24.txt
