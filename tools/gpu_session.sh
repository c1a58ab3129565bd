cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
kstat() {
  rm -rf /tmp/pr; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 30 > /dev/null 2>&1
  python - "$1" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/pr/**/*kernel_stats.csv', recursive=True)[0]
tot = 0
for r in csv.DictReader(open(f)):
    if 'k_rs_' in r['Name']:
        tot += float(r['TotalDurationNs']) / 33e3
        print(sys.argv[1], r['Name'][:28], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs']) / 1e3, 1))
print(sys.argv[1], 'sort per iteration us', round(tot, 1))
PY
}
{
kstat "in-tree t2048 d11:"
for v in t2048_d8 t1024_d8 t1024_d11 t512_d8; do ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_sort_$v.so kstat "$v:"; done
} > $O/s15_sort.txt 2>&1; cat $O/s15_sort.txt
timeout 600 python -m pytest tests/test_gpu_sort.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
for v in t1024_d8 t512_d8; do ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_sort_$v.so timeout 600 python -m pytest tests/test_gpu_sort.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -1; done
