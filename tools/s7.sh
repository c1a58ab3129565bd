cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
{
for rep in 1 2 3; do
  for r in 8 0 4 16; do
    if [ $r = 8 ]; then lib=""; else lib=$PWD/tools/ab_libs/libadfp_reserve$r.so; fi
    ADFP_LIB_PATH=$lib python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 300 --graph 2>/dev/null | tail -1 | sed "s/^/reserve $r, side lane ON, graph replay, 5000 x 64: /"
  done
  ADFP_SIDE_LANE=0 python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 300 --graph 2>/dev/null | tail -1 | sed "s/^/side lane OFF, graph replay, 5000 x 64: /"
done
} > $O/s7_reserve.txt 2>&1
cat $O/s7_reserve.txt
