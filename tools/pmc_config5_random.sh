cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_r04_c5r_fetch -- python3 tools/stress_config5.py random > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_r04_c5r_write -- python3 tools/stress_config5.py random > /dev/null 2>&1
python tools/pmc_traffic.py $O/pmc_r04_c5r_fetch $O/pmc_r04_c5r_write 16777216 $O/r04_pmc_hbm_config5_random.csv "tools/stress_config5.py random: 131072 rays x 128 samples, 1024^3 TSDF, each pose's rays drawn at random from its image"
python tools/stress_config5.py random | tail -1
python tools/stress_config5.py pixel | tail -1
rm -rf $O/pmc_r04_c5r_*
