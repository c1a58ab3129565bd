# Round-3 measurement collection (one gpurun call; everything lands in gpurun_out/, the summaries are copied to profiles/ by hand).
# rocprofv3: program directly after `--`; PMC passes separate from --stats passes, FETCH_SIZE and WRITE_SIZE in separate passes.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_r03_fetch -- python3 tools/ab_stage.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_r03_write -- python3 tools/ab_stage.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_r03_c5_fetch -- python3 tools/stress_config5.py pixel > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_r03_c5_write -- python3 tools/stress_config5.py pixel > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_r03_sq2 -- python3 tools/ab_stage.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_r03_sq1 -- python3 tools/ab_stage.py > /dev/null 2>&1
python tools/pmc_traffic.py $O/pmc_r03_fetch $O/pmc_r03_write 6400000 $O/r03_pmc_hbm_traffic.csv "tools/ab_stage.py: one 100000-ray batch" > /dev/null
python tools/pmc_traffic.py $O/pmc_r03_c5_fetch $O/pmc_r03_c5_write 16777216 $O/r03_pmc_hbm_config5.csv "tools/stress_config5.py pixel: 131072 rays x 128 samples, 1024^3 TSDF" > /dev/null
python tools/pmc_summary.py $O/pmc_r03_sq1 $O/pmc_r03_sq2 --match k_ > $O/r03_pmc_sq_forward.txt
cp $O/r03_pmc_hbm_traffic.csv $O/r03_pmc_hbm_config5.csv profiles/     # bench.py reads roofline.traffic from here: stamp = this build
python bench.py > $O/r03_bench_f16x3.json 2> $O/r03_bench_f16x3.err
ADFP_MATH=f32 python bench.py --cpu-rays 0 --no-extra > $O/r03_bench_f32.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03_bench -- python3 bench.py --cpu-rays 0 > /dev/null 2>&1
python profiles/summarize.py $O/prof_r03_bench $O/r03_kernel_stats_bench.csv > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03_headline -- python3 bench.py --cpu-rays 0 --no-extra --no-stage-timing > /dev/null 2>&1
python profiles/summarize.py $O/prof_r03_headline $O/r03_kernel_stats_headline.csv > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03_train -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked > /dev/null 2>&1
python profiles/summarize.py $O/prof_r03_train $O/r03_kernel_stats_train.csv > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_r03_train_fetch -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 10 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_r03_train_write -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 10 > /dev/null 2>&1
python tools/pmc_traffic.py $O/pmc_r03_train_fetch $O/pmc_r03_train_write 320000 $O/r03_pmc_hbm_train.csv "tools/profile_iteration.py --rays 5000 --samples 48 --masked: 320000 samples per iteration" > /dev/null
python tools/diag_bwd.py > $O/r03_diag_backward_modes.txt 2>/dev/null
python bench_train.py > $O/r03_bench_train.json 2>/dev/null
ADFP_MATH=f32 python bench_train.py --rays 5000 2>/dev/null | sed 's/^{/{"math": "f32 (exact forward + backward)", /' >> $O/r03_bench_train.json
python tools/mapping_loop.py --frames 200 --every-frame 5 2>/dev/null | tail -1 > $O/r03_mapping_loop.json
python tools/mapping_loop.py --frames 200 --every-frame 5 --fused 2>/dev/null | tail -1 >> $O/r03_mapping_loop.json
python tools/stress_config5.py pixel 2>/dev/null | tail -1 > $O/r03_config5.json
python tools/stress_config5.py random 2>/dev/null | tail -1 >> $O/r03_config5.json
# the raw rocprofv3 output directories are large (gpurun merges at most 64 MiB back): keep the summaries only
rm -rf $O/pmc_r03_* $O/prof_r03_*
wc -c $O/r03_bench_f16x3.json; tail -c 300 $O/r03_bench_f16x3.err; head -14 $O/r03_kernel_stats_headline.csv
