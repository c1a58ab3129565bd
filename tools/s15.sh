cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for st in low high color; do
for v in 0 low+high+color; do
ADFP_SIDE_LANE=$v python tools/profile_iteration.py --scene office0 --stage $st --rays 5000 --samples 48 --masked --iters 300 --graph 2>/dev/null | tail -1 | sed "s/^/office0 stage $st, ADFP_SIDE_LANE=$v: /"
done
done
done
