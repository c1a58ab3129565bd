# A - B - A - B of the fused Mapper iteration on one box: in-tree library (k_decode_lc16_train: low + colour training forward in ONE launch)
# against tools/ab_libs/libadfp_lc32.so (-DADFP_LC_32X32: k_decode_h<LOW, TRAIN> + k_decode_h<COLOR, TRAIN>), then the kernels' durations
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "5000 48" "1000 32"; do
  set -- $cfg
  for rep in 1 2; do
    a=$(python tools/profile_iteration.py --rays $1 --samples $2 --masked --iters 300 2>/dev/null | tail -1)
    b=$(ADFP_IMAGES=hg ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_lc32.so python tools/profile_iteration.py --rays $1 --samples $2 --masked --iters 300 2>/dev/null | tail -1)
    echo "rays $1 samples $2+16  one launch: $a   two launches: $b"
  done
done
for lib in "" $PWD/tools/ab_libs/libadfp_lc32.so; do
  if [ -n "$lib" ]; then export ADFP_LIB_PATH=$lib ADFP_IMAGES=hg; fi
  rm -rf /tmp/abt; (cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abt -- python3 $GRAFT_REPO_ROOT/tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 50 > /tmp/abt.log 2>&1)
  python - "${lib:-in-tree}" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/abt/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('k_decode_lc16_train', 'k_decode_h<32')):
        print('   ', sys.argv[1].split('/')[-1], r['Name'][:52], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs']) / 1e3, 1))
PY
done
