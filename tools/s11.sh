cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
{
for rep in 1 2 3; do
  python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 300 --graph 2>/dev/null | tail -1 | sed "s/^/in-tree (plain row stores)      : /"
  ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_nt_rows.so python tools/profile_iteration.py --rays 5000 --samples 48 --masked --iters 300 --graph 2>/dev/null | tail -1 | sed "s/^/non-temporal row stores (NT_ROWS): /"
done
python tools/ab_train_fwd.py 5000 48 300 2>/dev/null | tail -1
ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_nt_rows.so python tools/ab_train_fwd.py 5000 48 300 2>/dev/null | tail -1
} > $O/s11_nt_rows.txt 2>&1
cat $O/s11_nt_rows.txt
