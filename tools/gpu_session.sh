cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_shared_params.py tests/test_gpu_tsdf_blocks.py tests/test_gpu_config5.py tests/test_gpu_latch_order.py -m gpu -q -p no:cacheprovider > $O/s3_pytest.log 2>&1
echo "pytest rc=$?" >> $O/s3_pytest.log; tail -5 $O/s3_pytest.log
{
echo "# per-role time of k_decode_bwd_roles (timing-only builds in which only one role's workgroups run), 5000 x 64, kernel-trace"
for r in 0 1 2; do
  rm -rf /tmp/pr; ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_role$r.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 30 > /dev/null 2>&1
  python - $r <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/pr/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_decode_bwd_roles' in r['Name']:
        print('only role', sys.argv[1], r['Name'][:50], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs']) / 1e3, 1), 'min', round(float(r['MinNs']) / 1e3, 1))
PY
done
echo "# training forward alone (HIP events), in-tree vs no X rows at all vs no head + c pieces"
for rep in 1 2; do
  for lib in "" tools/ab_libs/libadfp_train_NOX.so tools/ab_libs/libadfp_train_NOHEADC.so; do
    if [ -n "$lib" ]; then export ADFP_LIB_PATH=$PWD/$lib; else unset ADFP_LIB_PATH; fi
    timeout 200 python tools/ab_train_fwd.py 5000 48 2>&1 | tail -1
  done
done
unset ADFP_LIB_PATH
for lib in "" tools/ab_libs/libadfp_train_NOX.so tools/ab_libs/libadfp_train_NOHEADC.so; do
  if [ -n "$lib" ]; then export ADFP_LIB_PATH=$PWD/$lib; else unset ADFP_LIB_PATH; fi
  rm -rf /tmp/pr; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -- python3 tools/ab_train_fwd.py 5000 48 50 > /dev/null 2>&1
  python - "${lib:-in-tree}" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/pr/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('k_decode_lc16_train', 'k_attention_h', 'k_decode_h<64')):
        print('   ', sys.argv[1].split('/')[-1], r['Name'][:52], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs']) / 1e3, 1))
PY
done
unset ADFP_LIB_PATH
} > $O/s3_ab.txt 2>&1
cat $O/s3_ab.txt
# in-band bisect: the in-tree high decoder vs the LOW network's body in its launch shape
for lib in "" tools/ab_libs/libadfp_high_as_low.so; do
  tag=intree; if [ -n "$lib" ]; then export ADFP_LIB_PATH=$PWD/$lib; tag=high_as_low; else unset ADFP_LIB_PATH; fi
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_ib1_$tag -- python3 tools/ab_stage.py > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_WAIT_ANY --output-format csv -d $O/pmc_ib2_$tag -- python3 tools/ab_stage.py > /dev/null 2>&1
  echo "#### $tag"; python tools/pmc_summary.py $O/pmc_ib1_$tag $O/pmc_ib2_$tag --match k_decode_high_g
  echo "#### $tag (k_decode_lc16 for reference)"; python tools/pmc_summary.py $O/pmc_ib1_$tag $O/pmc_ib2_$tag --match k_decode_lc16
  rm -rf $O/pmc_ib1_$tag $O/pmc_ib2_$tag
done > $O/s3_inband_bisect.txt 2>&1
unset ADFP_LIB_PATH
tail -60 $O/s3_inband_bisect.txt
