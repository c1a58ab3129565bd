# The headline on THIS lease with the in-tree library and with tools/ab_libs/libadfp_lc32.so (-DADFP_LC_32X32: the round-3 32x32x16
# kernels, fixed tile split), driver flags; one line per library.  Run on several leases: boxes differ by +-6 %, the ratio does not.
cd $GRAFT_REPO_ROOT
for lib in "" lc32 "" lc32; do
  if [ -n "$lib" ]; then export ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_$lib.so ADFP_IMAGES=hg; else unset ADFP_LIB_PATH ADFP_IMAGES; fi
  python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-rays 0 --no-extra --no-stage-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${lib:-round-4 kernels}', round(d['value']/1e6,2), 'M rays/s', round(d['ms_per_step'],3), 'ms per frame')"
done
