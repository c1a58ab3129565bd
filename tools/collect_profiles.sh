# Round-4 measurement collection (one gpurun call; everything lands in gpurun_out/, the summaries are copied to profiles/ by hand).
# rocprofv3: program directly after `--`; PMC passes separate from --stats passes, FETCH_SIZE and WRITE_SIZE in separate passes.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
R=r04
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${R}_fetch -- python3 tools/ab_stage.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${R}_write -- python3 tools/ab_stage.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${R}_c5_fetch -- python3 tools/stress_config5.py pixel > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${R}_c5_write -- python3 tools/stress_config5.py pixel > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_${R}_sq2 -- python3 tools/ab_stage.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${R}_sq1 -- python3 tools/ab_stage.py > /dev/null 2>&1
python tools/pmc_traffic.py $O/pmc_${R}_fetch $O/pmc_${R}_write 6400000 $O/${R}_pmc_hbm_traffic.csv "tools/ab_stage.py: one 100000-ray batch" > /dev/null
python tools/pmc_traffic.py $O/pmc_${R}_c5_fetch $O/pmc_${R}_c5_write 16777216 $O/${R}_pmc_hbm_config5.csv "tools/stress_config5.py pixel: 131072 rays x 128 samples, 1024^3 TSDF" > /dev/null
python tools/pmc_summary.py $O/pmc_${R}_sq1 $O/pmc_${R}_sq2 --match k_ > $O/${R}_pmc_sq_forward.txt
cp $O/${R}_pmc_hbm_traffic.csv $O/${R}_pmc_hbm_config5.csv $O/${R}_pmc_sq_forward.txt profiles/     # bench.py reads roofline.traffic / limiter.pmc from here: stamp = this build
python bench.py > $O/${R}_bench_f16x3.json 2> $O/${R}_bench_f16x3.err
ADFP_MATH=f32 python bench.py --cpu-rays 0 --no-extra > $O/${R}_bench_f32.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${R}_bench -- python3 bench.py --cpu-rays 0 > /dev/null 2>&1
python profiles/summarize.py $O/prof_${R}_bench $O/${R}_kernel_stats_bench.csv > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${R}_headline -- python3 bench.py --cpu-rays 0 --no-extra --no-stage-timing > /dev/null 2>&1
python profiles/summarize.py $O/prof_${R}_headline $O/${R}_kernel_stats_headline.csv > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${R}_train -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked > /dev/null 2>&1
python profiles/summarize.py $O/prof_${R}_train $O/${R}_kernel_stats_train.csv > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${R}_unch -- python3 tools/host_breakdown.py --rays 5000 --no-tracker --iters 100 > /dev/null 2>&1
python profiles/summarize.py $O/prof_${R}_unch $O/${R}_kernel_stats_unchanged_5000x64.csv > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${R}_train_fetch -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 10 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${R}_train_write -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 10 > /dev/null 2>&1
python tools/pmc_traffic.py $O/pmc_${R}_train_fetch $O/pmc_${R}_train_write 320000 $O/${R}_pmc_hbm_train.csv "tools/profile_iteration.py --rays 5000 --samples 48 --masked: 320000 samples per iteration" > /dev/null
python tools/diag_bwd.py > $O/${R}_diag_backward_modes.txt 2>/dev/null
python bench_train.py > $O/${R}_bench_train.json 2>/dev/null
ADFP_MATH=f32 python bench_train.py --rays 5000 2>/dev/null | sed 's/^{/{"math": "f32 (exact forward + backward)", /' >> $O/${R}_bench_train.json
ADFP_HOST_TIMING=1 python tools/host_breakdown.py --rays 1000 5000 > $O/${R}_host_breakdown_box2.txt 2>&1   # (r04_host_breakdown.txt = the lease DESIGN section 4.6 quotes)
python tools/mapping_loop.py --frames 200 --every-frame 5 2>/dev/null | tail -1 > $O/${R}_mapping_loop.json
python tools/mapping_loop.py --frames 200 --every-frame 5 --fused 2>/dev/null | tail -1 >> $O/${R}_mapping_loop.json
python tools/stress_config5.py pixel 2>/dev/null | tail -1 > $O/${R}_config5.json
python tools/stress_config5.py random 2>/dev/null | tail -1 >> $O/${R}_config5.json
bash tools/ab_lc.sh > $O/${R}_ab_lc16.txt 2>/dev/null
(cd tools/micro && ./mfma_shape_ab > ../../$O/${R}_micro_mfma_shape_ab.txt 2>/dev/null)
hostname > $O/${R}_box.txt; rocm-smi --showproductname 2>/dev/null | head -8 >> $O/${R}_box.txt
# the raw rocprofv3 output directories are large (gpurun merges at most 64 MiB back): keep the summaries only
rm -rf $O/pmc_${R}_* $O/prof_${R}_*
wc -c $O/${R}_bench_f16x3.json; tail -c 300 $O/${R}_bench_f16x3.err; head -14 $O/${R}_kernel_stats_headline.csv
