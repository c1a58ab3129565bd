# Round-6 measurement collection (one gpurun call; everything lands in gpurun_out/, the summaries are copied to profiles/ by hand
# apart from the three PMC files bench.py reads, which are copied here so that the bench lines carry this build's traffic).
#
#   bash tools/collect_profiles.sh [section ...]        sections: tests pmc bench shard train diag config5 loops ab    (default: all)
#
# Every command's stderr is kept (gpurun_out/r05_logs/<step>.err) and a step that exits non-zero is reported, its target file is
# moved to the logs (<step>.failed_output: no empty or half-written summary is left to be copied), and the script itself exits 1 at the end.
# rocprofv3: program directly after `--`; PMC passes separate from --stats passes, FETCH_SIZE and WRITE_SIZE in separate passes.
set -o pipefail
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
R=r06
L=$O/${R}_logs
mkdir -p $L
FAILED=""
SECTIONS="${*:-tests pmc bench shard train diag config5 loops ab}"
want() { case " $SECTIONS " in *" $1 "*) return 0;; esac; return 1; }

# run <step> <command ...>: stdout -> logs/<step>.out
run() {
  local step=$1; shift
  "$@" > $L/$step.out 2> $L/$step.err
  local rc=$?
  if [ $rc -ne 0 ]; then FAILED="$FAILED $step"; echo "FAILED rc=$rc: $step: $*"; tail -n 8 $L/$step.err; fi
  return $rc
}
# into <file> <step> <command ...>: stdout -> <file>; the file is removed when the command fails
into() {
  local file=$1 step=$2; shift 2
  "$@" > $file 2> $L/$step.err
  local rc=$?
  if [ $rc -ne 0 ]; then FAILED="$FAILED $step"; echo "FAILED rc=$rc: $step: $*"; tail -n 8 $L/$step.err; mv -f $file $L/$step.failed_output 2> /dev/null; tail -n 15 $L/$step.failed_output; fi
  return $rc
}
prof_stats() {   # <dir> <step> <program ...>: rocprofv3 --kernel-trace --stats
  local d=$1 step=$2; shift 2
  rm -rf $d; run $step rocprofv3 --kernel-trace --stats --output-format csv -d $d -- "$@"
}
prof_pmc() {     # <dir> <step> "<counters>" <program ...>
  local d=$1 step=$2 ctr=$3; shift 3
  rm -rf $d; run $step rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $d -- "$@"
}

if want pmc; then
  prof_pmc $O/pmc_fetch pmc_fetch FETCH_SIZE python3 tools/ab_stage.py
  prof_pmc $O/pmc_write pmc_write WRITE_SIZE python3 tools/ab_stage.py
  run pmc_traffic python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 6400000 $O/${R}_pmc_hbm_traffic.csv "tools/ab_stage.py: one 100000-ray batch"
  prof_pmc $O/pmc_c5_fetch pmc_c5_fetch FETCH_SIZE python3 tools/stress_config5.py pixel
  prof_pmc $O/pmc_c5_write pmc_c5_write WRITE_SIZE python3 tools/stress_config5.py pixel
  run pmc_c5_traffic python tools/pmc_traffic.py $O/pmc_c5_fetch $O/pmc_c5_write 16777216 $O/${R}_pmc_hbm_config5.csv "tools/stress_config5.py pixel: 131072 rays x 128 samples, 1024^3 TSDF"
  prof_pmc $O/pmc_c5r_fetch pmc_c5r_fetch FETCH_SIZE python3 tools/stress_config5.py random
  prof_pmc $O/pmc_c5r_write pmc_c5r_write WRITE_SIZE python3 tools/stress_config5.py random
  run pmc_c5r_traffic python tools/pmc_traffic.py $O/pmc_c5r_fetch $O/pmc_c5r_write 16777216 $O/${R}_pmc_hbm_config5_random.csv "tools/stress_config5.py random: 131072 rays x 128 samples, 1024^3 TSDF, each pose's rays drawn at random from its image; the renderer's own choice (corner blocks, no sort)"
  prof_pmc $O/pmc_sq1 pmc_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" python3 tools/ab_stage.py
  prof_pmc $O/pmc_sq2 pmc_sq2 "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" python3 tools/ab_stage.py
  into $O/${R}_pmc_sq_forward.txt pmc_sq_summary python tools/pmc_summary.py $O/pmc_sq1 $O/pmc_sq2 --match k_
  prof_pmc $O/pmc_train_fetch pmc_train_fetch FETCH_SIZE python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 10
  prof_pmc $O/pmc_train_write pmc_train_write WRITE_SIZE python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 10
  run pmc_train_traffic python tools/pmc_traffic.py $O/pmc_train_fetch $O/pmc_train_write 320000 $O/${R}_pmc_hbm_train.csv "tools/profile_iteration.py --rays 5000 --samples 48 --masked: 320000 samples per iteration"
  # bench.py reads roofline.traffic / limiter.pmc from profiles/: the stamp inside each file says which build it belongs to
  for f in ${R}_pmc_hbm_traffic.csv ${R}_pmc_hbm_config5.csv ${R}_pmc_hbm_config5_random.csv ${R}_pmc_sq_forward.txt; do [ -s $O/$f ] && cp $O/$f profiles/; done
fi

if want bench; then
  into $O/${R}_bench_f16x3.json bench_f16x3 python bench.py
  ADFP_MATH=f32 into $O/${R}_bench_f32.json bench_f32 python bench.py --cpu-rays 0 --no-extra
  prof_stats $O/prof_bench prof_bench python3 bench.py --cpu-rays 0
  run sum_bench python profiles/summarize.py $O/prof_bench $O/${R}_kernel_stats_bench.csv
  prof_stats $O/prof_headline prof_headline python3 bench.py --cpu-rays 0 --no-extra --no-stage-timing --no-shard-model
  run sum_headline python profiles/summarize.py $O/prof_headline $O/${R}_kernel_stats_headline.csv
fi

if want shard; then
  # one rank's 1/8 share of the ray-sharded headline frame, run the way bench.py --gpus 8 runs it
  into $O/${R}_shard_pipelined.txt shard_pipelined python tools/shard_step.py
  prof_stats $O/prof_shard8 prof_shard8 python3 tools/shard_step.py --trace 8 3
  run sum_shard8 python profiles/summarize.py $O/prof_shard8 $O/${R}_kernel_stats_shard8.csv
  f=$(find $O/prof_shard8 -name '*kernel_trace.csv' | head -1)
  into $O/${R}_shard8_timeline.txt shard8_timeline python tools/trace_timeline.py $f k_forward_head
  into $O/${R}_weight_histogram.txt weight_histogram python tools/weight_histogram.py
fi

if want train; then
  prof_stats $O/prof_train prof_train python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked
  run sum_train python profiles/summarize.py $O/prof_train $O/${R}_kernel_stats_train.csv
  into $O/${R}_bench_train.json bench_train python bench_train.py
  ADFP_MATH=f32 into $O/${R}_bench_train_f32.json bench_train_f32 python bench_train.py --rays 5000
  {
    for rep in 1 2; do
      python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 200 --graph | tail -1 | sed "s/^/fused iteration, graph replay, 5000 x 64: /"
      python tools/profile_iteration.py --rays 1000 --samples 32 --masked --iters 200 --graph | tail -1 | sed "s/^/fused iteration, graph replay, 1000 x 48: /"
      ADFP_SIDE_LANE=0 python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 200 --graph | tail -1 | sed "s/^/one stream (ADFP_SIDE_LANE=0), graph replay, 5000 x 64: /"
    done
  } > $O/${R}_fused_iteration.txt 2> $L/fused_iteration.err || FAILED="$FAILED fused_iteration"
fi

if want tests; then
  # the whole GPU suite, with every comparison logged: the pass / fail line, r05_parity_stats.txt, r05_grad_stats.txt
  rm -f $O/${R}_parity_raw.txt $O/${R}_grad_raw.txt
  ADFP_PARITY_STATS=$PWD/$O/${R}_parity_raw.txt ADFP_GRAD_STATS=$PWD/$O/${R}_grad_raw.txt into $O/${R}_pytest_gpu.txt pytest_gpu timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider
  tail -n 3 $O/${R}_pytest_gpu.txt
  into $O/${R}_parity_stats.txt parity_stats python tools/stats_summary.py parity $O/${R}_parity_raw.txt
  into $O/${R}_grad_stats.txt grad_stats python tools/stats_summary.py grad $O/${R}_grad_raw.txt
  rm -f $O/${R}_parity_raw.txt $O/${R}_grad_raw.txt
fi

if want diag; then
  into $O/${R}_diag_backward_modes.txt diag_bwd python tools/diag_bwd.py
fi

if want config5; then
  into $O/${R}_config5.json c5_pixel python tools/stress_config5.py pixel
  for order in random random-sorted; do
    python tools/stress_config5.py $order 2> $L/c5_$order.err | tail -1 >> $O/${R}_config5.json || FAILED="$FAILED c5_$order"
  done
fi

if want loops; then
  into $O/${R}_mapping_loop.json loop_plain python tools/mapping_loop.py --frames 200 --every-frame 5
  python tools/mapping_loop.py --frames 200 --every-frame 5 --fused 2> $L/loop_fused.err | tail -1 >> $O/${R}_mapping_loop.json || FAILED="$FAILED loop_fused"
  ADFP_HOST_TIMING=1 into $O/${R}_host_breakdown.txt host_breakdown python tools/host_breakdown.py --rays 1000 5000
  [ -d tools/ab_r05 ] && into $O/${R}_host_ab.txt host_ab bash tools/host_ab.sh
fi

if want ab; then
  into $O/${R}_ab_fourier_mfma.txt ab_fourier_mfma bash tools/ab_fourier_mfma.sh
  into $O/${R}_ab_backward_roles.txt ab_roles bash tools/ab_roles.sh
  ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_roles_span.so python tools/roles_span.py 2> $L/roles_span.err | tail -4 >> $O/${R}_ab_backward_roles.txt || FAILED="$FAILED roles_span"
  {
    for lib in "" train_NOX train_NOC; do
      for rep in 1 2; do
        if [ -n "$lib" ]; then ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_$lib.so python tools/ab_train_fwd.py 5000 48 300; else python tools/ab_train_fwd.py 5000 48 300; fi
      done
    done
  } > $O/${R}_ab_train_forward.txt 2> $L/ab_train_forward.err || FAILED="$FAILED ab_train_forward"
fi

{ hostname; rocm-smi --showproductname 2> $L/box.err | head -8; } > $O/${R}_box.txt
# the raw rocprofv3 output directories are large (gpurun merges at most 64 MiB back): keep the summaries only
rm -rf $O/pmc_* $O/prof_*
ls -la $O | grep ${R}_ | awk '{print $5, $9}'
if [ -n "$FAILED" ]; then echo "collect_profiles: FAILED steps:$FAILED"; exit 1; fi
echo "collect_profiles: all steps ended with exit code 0"
