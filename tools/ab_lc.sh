# A - B - A of the fused low + colour launch on one box: the in-tree library (16x16x32) against tools/ab_libs/libadfp_lc32.so (the 32x32x16 form)
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  AB_REPS=12 python tools/ab_stage.py 2>/dev/null | tail -1
  AB_REPS=12 ADFP_IMAGES=hg ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_lc32.so python tools/ab_stage.py 2>/dev/null | tail -1
done
AB_REPS=12 python tools/ab_stage.py 2>/dev/null | tail -1
