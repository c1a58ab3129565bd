cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_grad.py tests/test_gpu_mapper_iteration.py tests/test_gpu_config3.py tests/test_gpu_nccl.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_x -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked > /dev/null 2>&1; python profiles/summarize.py gpurun_out/prof_x gpurun_out/x_train.csv > /dev/null; grep -E "reduce_partials|CatArray|k_outer_h|k_scatter" gpurun_out/x_train.csv | cut -c1-120
python tools/profile_iteration.py --rays 5000 --samples 48 --masked 2>/dev/null | tail -1; python tools/profile_iteration.py --rays 1000 --samples 32 --masked 2>/dev/null | tail -1
