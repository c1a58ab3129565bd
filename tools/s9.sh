cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
for rep in 1 2; do
  ADFP_HOST_TIMING=1 python tools/ab_r05/tools/host_breakdown.py --rays 5000 --no-tracker 2>/dev/null | grep -v "0.0 calls" | sed "s/^/r05 | /" 
  ADFP_HOST_TIMING=1 python tools/host_breakdown.py --rays 5000 --no-tracker 2>/dev/null | grep -v "0.0 calls" | sed "s/^/r06 | /"
done > $O/s9_host_ab.txt 2>&1
cat $O/s9_host_ab.txt
python tools/host_breakdown.py --rays 5000 --no-tracker --cprofile > $O/s9_cprofile.txt 2>&1
awk '/Ordered by: internal/{f=1} f' $O/s9_cprofile.txt | head -60 | cut -c1-150
