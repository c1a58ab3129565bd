# The backward's side lane (adfp_backward_args.side_stream), A/B on one lease -> profiles/r06_ab_side_lane.txt.
#   1. fused Mapper iteration, room0, 5 000 x 64, stage colour, graph replay: one stream (ADFP_SIDE_LANE=0) against the lane with
#      ADFP_SIDE_CU_RESERVE = 4 (in-tree) / 0 / 8 / 16 (tools/ab_libs/libadfp_reserve<n>.so: build with
#      hipcc <csrc/build.sh's flags> -DADFP_SIDE_CU_RESERVE=<n> -o tools/ab_libs/libadfp_reserve<n>.so adfp_kernels.hip; skipped when absent)
#   2. per stage at office0, one stream against the lane in every stage (ADFP_SIDE_LANE=low+high+color)
#   3. the office0 mapping loop under the four policies
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 300 --graph 2>/dev/null | tail -1 | sed "s/^/reserve 4 (in-tree), side lane ON, graph replay, 5000 x 64: /"
  for r in 0 8 16; do
    lib=$PWD/tools/ab_libs/libadfp_reserve$r.so
    [ -f $lib ] && ADFP_LIB_PATH=$lib python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 300 --graph 2>/dev/null | tail -1 | sed "s/^/reserve $r, side lane ON, graph replay, 5000 x 64: /"
  done
  ADFP_SIDE_LANE=0 python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 300 --graph 2>/dev/null | tail -1 | sed "s/^/side lane OFF, graph replay, 5000 x 64: /"
done
for rep in 1 2; do
  for st in low high color; do
    for v in 0 low+high+color; do
      ADFP_SIDE_LANE=$v python tools/profile_iteration.py --scene office0 --stage $st --rays 5000 --samples 48 --masked --iters 300 --graph 2>/dev/null | tail -1 | sed "s/^/office0 stage $st, ADFP_SIDE_LANE=$v: /"
    done
  done
done
for rep in 1 2 3; do
  for v in 0 color high+color low+high+color; do
    ADFP_SIDE_LANE=$v python tools/mapping_loop.py --frames 200 --every-frame 5 --fused 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('office0 mapping loop, fused, ADFP_SIDE_LANE=%-16s ms per iteration %.4f' % ('$v', r['ms_per_iteration']))"
  done
done
