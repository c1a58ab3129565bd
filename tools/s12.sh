cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/s12_pytest.txt 2>&1; grep -n "passed\|failed" $O/s12_pytest.txt; grep -n "Error\|assert" $O/s12_pytest.txt | head -20
for rep in 1 2 3; do
  python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 300 --graph 2>/dev/null | tail -1 | sed "s/^/5000 x 64: /"
  python tools/profile_iteration.py --rays 1000 --samples 32 --masked --iters 300 --graph 2>/dev/null | tail -1 | sed "s/^/1000 x 48: /"
done
python tools/mapping_loop.py --frames 200 --every-frame 5 --fused 2>/dev/null | tail -1 | cut -c1-300
