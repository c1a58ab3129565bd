cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
python tools/host_breakdown.py --rays 1000 --no-tracker --cprofile > $O/s29_cprofile.txt 2>&1
tail -90 $O/s29_cprofile.txt | cut -c1-150
