# A - B - A of the in-band pair on one box: in-tree library against a build given as $1 (e.g. tools/ab_libs/libadfp_high768.so);
# kernel durations of one 100 000-ray batch from rocprofv3 --kernel-trace --stats, and the unprofiled batch wall time
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B=$PWD/$1
run() {   # $1 = label, $2 = lib path or empty
  if [ -n "$2" ]; then export ADFP_LIB_PATH=$2; else unset ADFP_LIB_PATH; fi
  AB_REPS=12 python tools/ab_stage.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['lib'], d['checksum'], 'batch_ms', d['batch_ms'], 'low_color_ms', d.get('low_color_ms'))"
  rm -rf /tmp/abh; cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abh -- python3 $GRAFT_REPO_ROOT/tools/ab_stage.py > /tmp/abh.log 2>&1 || tail -5 /tmp/abh.log; cd $GRAFT_REPO_ROOT
  python - "$1" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/abh/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('k_decode_high', 'k_attention', 'k_decode_h<64')):
        print('   ', sys.argv[1], r['Name'][:48], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs']) / 1e3, 1), 'min_us', round(float(r['MinNs']) / 1e3, 1))
PY
}
run A ""
run B $B
run A ""
run B $B
