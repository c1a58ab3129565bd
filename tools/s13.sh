cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
python tools/mapping_loop.py --frames 200 --every-frame 5 --fused 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('side lane on :', r['ms_per_iteration'], r['seconds'])"
ADFP_SIDE_LANE=0 python tools/mapping_loop.py --frames 200 --every-frame 5 --fused 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('side lane off:', r['ms_per_iteration'], r['seconds'])"
done
python -m cProfile -s tottime tools/mapping_loop.py --frames 200 --every-frame 5 --fused 2>/dev/null | head -45 | cut -c1-150
