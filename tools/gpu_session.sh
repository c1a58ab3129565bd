cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_tracker_iteration.py tests/test_gpu_config3.py tests/test_gpu_mapper_iteration.py tests/test_gpu_errors.py tests/test_gpu_abi_demo.py -m gpu -q -p no:cacheprovider -x > $O/s18_pytest.log 2>&1
echo "pytest rc=$?" >> $O/s18_pytest.log; tail -5 $O/s18_pytest.log
python tools/mapping_loop.py --frames 200 --every-frame 5 --fused 2>&1 | tail -1 | cut -c1-400
python tools/mapping_loop.py --frames 200 --every-frame 5 --fused 2>&1 | tail -1 | cut -c1-400
