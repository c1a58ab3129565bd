cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in 0 color high+color low+high+color; do
ADFP_SIDE_LANE=$v python tools/mapping_loop.py --frames 200 --every-frame 5 --fused 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('office0 mapping loop, fused, ADFP_SIDE_LANE=%-16s ms per iteration %.4f' % ('$v', r['ms_per_iteration']))"
done
done
