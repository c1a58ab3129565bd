# Per-role time of k_decode_bwd_roles: timing-only builds in which only ONE role's workgroups run (-DADFP_EXP_ONLY_ROLE=0/1/2 ->
# tools/ab_libs/libadfp_role<r>.so), each at its share of the workgroups, on the 5 000-ray x 64-sample Mapper iteration; then the
# in-tree kernel and round 3-4's one-wave kernel (ADFP_WGRAD=o) on the same lease.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
kstat() {   # $1 = label; kernel-trace of the iteration, prints the two big training kernels
  rm -rf /tmp/pr; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 30 > /dev/null 2>&1
  python - "$1" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/pr/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('k_decode_bwd_roles', 'k_decode_bwd_fused', 'k_decode_lc16_train')):
        print(sys.argv[1], r['Name'][:48], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs']) / 1e3, 1), 'min', round(float(r['MinNs']) / 1e3, 1))
PY
}
for r in 0 1 2; do ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_role$r.so kstat "only role $r:"; done
kstat "in-tree:"
ADFP_WGRAD=o kstat "one wave per SIMD (round 4):"
for rep in 1 2; do
  python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 200 --graph 2>&1 | tail -1 | sed "s/^/role-split  graph 5000x64: /"
  ADFP_WGRAD=o python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 200 --graph 2>&1 | tail -1 | sed "s/^/one-wave    graph 5000x64: /"
  python tools/profile_iteration.py --rays 1000 --samples 32 --masked --iters 200 --graph 2>&1 | tail -1 | sed "s/^/role-split  graph 1000x48: /"
  ADFP_WGRAD=o python tools/profile_iteration.py --rays 1000 --samples 32 --masked --iters 200 --graph 2>&1 | tail -1 | sed "s/^/one-wave    graph 1000x48: /"
done
