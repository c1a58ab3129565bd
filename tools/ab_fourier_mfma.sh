# A - B - A on one box: the in-tree library (p @ B of the Fourier features as an fma chain on the VALU) against
# tools/ab_libs/libadfp_pb_mfma.so (-DADFP_PB_MFMA: the same products on v_mfma_f32_32x32x2_f32).  Same lease; per line: the
# sha256 of depth / uncertainty / colour / weight / raw / z_vals of one 100 000-ray batch (bit-identity), the fused low + colour
# launch (median, min of 12), the whole render_batch_ray, and below the headline frame through bench.py.
cd $GRAFT_REPO_ROOT
show() { python -c "
import json,sys
r=json.loads(sys.stdin.readline())
print('%-22s sha %s  low+colour launch %.4f / %.4f ms  colour alone %.4f  whole batch %.4f ms' % (r['lib'], r['sha256_of_outputs_raw_z'], r['low_color_ms'][0], r['low_color_ms'][1], r['color_ms'][0], r['batch_ms'][0]))"; }
for round in 1 2; do
  AB_REPS=12 python tools/ab_stage.py 2>/dev/null | tail -1 | show
  AB_REPS=12 ADFP_LIB_PATH=$PWD/tools/ab_libs/libadfp_pb_mfma.so python tools/ab_stage.py 2>/dev/null | tail -1 | show
done
AB_REPS=12 python tools/ab_stage.py 2>/dev/null | tail -1 | show
for lib in "" $PWD/tools/ab_libs/libadfp_pb_mfma.so ""; do
  ADFP_LIB_PATH=$lib python bench.py --steps 20 --warmup 5 --cpu-rays 0 --no-extra --no-stage-timing --no-shard-model 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('bench.py headline, lib = %-40s %.3f ms per frame, %.2f M rays/s' % ('$lib' or 'in-tree', r['ms_per_step'], r['value']/1e6))"
done
