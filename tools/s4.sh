# round-6 session 4: the backward's side lane (sort beside the backward kernels): tests, fused iteration with and without it, a trace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/s4_pytest.txt 2>&1; tail -3 $O/s4_pytest.txt
{
for rep in 1 2; do
  python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 200 --graph | tail -1 | sed "s/^/side lane ON , graph replay, 5000 x 64: /"
  ADFP_SIDE_LANE=0 python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 200 --graph | tail -1 | sed "s/^/side lane OFF, graph replay, 5000 x 64: /"
  python tools/profile_iteration.py --rays 1000 --samples 32 --masked --iters 200 --graph | tail -1 | sed "s/^/side lane ON , graph replay, 1000 x 48: /"
  ADFP_SIDE_LANE=0 python tools/profile_iteration.py --rays 1000 --samples 32 --masked --iters 200 --graph | tail -1 | sed "s/^/side lane OFF, graph replay, 1000 x 48: /"
done
python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 200 | tail -1 | sed "s/^/side lane ON , eager, 5000 x 64: /"
ADFP_SIDE_LANE=0 python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 200 | tail -1 | sed "s/^/side lane OFF, eager, 5000 x 64: /"
} > $O/s4_side_lane.txt 2>&1
cat $O/s4_side_lane.txt
rm -rf $O/prof_it
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_it -- python3 tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 30 > $O/s4_trace.log 2>&1
f=$(find $O/prof_it -name '*kernel_trace.csv' | head -1)
python tools/trace_timeline.py $f k_prefilter_mask > $O/s4_iter_timeline.txt 2>&1; cat $O/s4_iter_timeline.txt
rm -rf $O/prof_it
