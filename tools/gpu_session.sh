cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
kstat() {
  rm -rf /tmp/pr; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 30 > /dev/null 2>&1
  python - "$1" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/pr/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_decode_bwd_roles' in r['Name']:
        print(sys.argv[1], r['Name'][:40], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs']) / 1e3, 1), 'min', round(float(r['MinNs']) / 1e3, 1))
PY
}
{
for sh in 98,98 103,86 106,84 100,88 104,82 102,90 108,80 100,84 104,88; do ADFP_ROLE_SHARES=$sh kstat "shares $sh:"; done
} > $O/s12_roles.txt 2>&1; cat $O/s12_roles.txt
for sh in 98,98 103,86 104,84; do
  ADFP_ROLE_SHARES=$sh python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 200 --graph 2>&1 | tail -1 | sed "s/^/shares $sh graph 5000x64: /"
  ADFP_ROLE_SHARES=$sh python tools/profile_iteration.py --rays 1000 --samples 32 --masked --iters 200 --graph 2>&1 | tail -1 | sed "s/^/shares $sh graph 1000x48: /"
done > $O/s12_iter.txt 2>&1; cat $O/s12_iter.txt
