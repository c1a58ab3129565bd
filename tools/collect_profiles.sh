cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_r02_fetch -- python3 tools/ab_stage.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_r02_write -- python3 tools/ab_stage.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_r02_c5_fetch -- python3 tools/stress_config5.py pixel > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_r02_c5_write -- python3 tools/stress_config5.py pixel > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_r02_sq2 -- python3 tools/ab_stage.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_r02_sq1 -- python3 tools/ab_stage.py > /dev/null 2>&1
python tools/pmc_traffic.py $O/pmc_r02_fetch $O/pmc_r02_write 6400000 $O/r02_pmc_hbm_traffic.csv "tools/ab_stage.py: one 100000-ray batch" > /dev/null
python tools/pmc_traffic.py $O/pmc_r02_c5_fetch $O/pmc_r02_c5_write 16777216 $O/r02_pmc_hbm_config5.csv "tools/stress_config5.py pixel: 131072 rays x 128 samples, 1024^3 TSDF" > /dev/null
python tools/pmc_summary.py $O/pmc_r02_sq1 $O/pmc_r02_sq2 --match k_ > $O/r02_pmc_sq_forward.txt
cp $O/r02_pmc_hbm_traffic.csv $O/r02_pmc_hbm_config5.csv profiles/     # bench.py reads roofline.traffic from here: stamp = this build
python bench.py > $O/r02_bench_f16x3.json 2> $O/r02_bench_f16x3.err
ADFP_MATH=f32 python bench.py --cpu-rays 0 --no-extra > $O/r02_bench_f32.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r02_bench -- python3 bench.py --cpu-rays 0 > /dev/null 2>&1
python profiles/summarize.py $O/prof_r02_bench $O/r02_kernel_stats_bench.csv > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r02_train -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked > /dev/null 2>&1
python profiles/summarize.py $O/prof_r02_train $O/r02_kernel_stats_train.csv > /dev/null
ADFP_MATH=f32 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r02_train_f32 -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked > /dev/null 2>&1
python profiles/summarize.py $O/prof_r02_train_f32 $O/r02_kernel_stats_train_exact_backward.csv > /dev/null
python tools/diag_bwd.py > $O/r02_diag_backward_modes.txt 2>/dev/null
python bench_train.py > $O/r02_bench_train.json 2>/dev/null
ADFP_MATH=f32 python bench_train.py --rays 5000 2>/dev/null | sed 's/^{/{"math": "f32 (exact forward + backward)", /' >> $O/r02_bench_train.json
python tools/mapping_loop.py --frames 20 2>/dev/null | tail -1 > $O/r02_mapping_loop.json
python tools/mapping_loop.py --frames 20 --fused 2>/dev/null | tail -1 >> $O/r02_mapping_loop.json
python tools/stress_config5.py pixel 2>/dev/null | tail -1 > $O/r02_config5.json
python tools/stress_config5.py random 2>/dev/null | tail -1 >> $O/r02_config5.json
wc -c $O/r02_bench_f16x3.json; tail -c 300 $O/r02_bench_f16x3.err; head -12 $O/r02_kernel_stats_bench.csv
