cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for d in 0 1 2 4 8 15; do echo "dbg $d: $(ADFP_DBG_SCATTER=$d python tools/profile_iteration.py --rays 5000 --samples 48 --masked 2>/dev/null | tail -1)"; done
