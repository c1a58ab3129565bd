cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > $O/r04_t1.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r04_unch -- python3 tools/host_breakdown.py --rays 5000 --no-tracker --iters 100 > $O/r04_unch_stdout.txt 2>&1
python profiles/summarize.py $O/prof_r04_unch $O/r04_kernel_stats_unchanged_5000x64_step1.csv > /dev/null
rm -rf $O/prof_r04_unch
cat $O/r04_t1.txt; head -60 $O/r04_kernel_stats_unchanged_5000x64_step1.csv
