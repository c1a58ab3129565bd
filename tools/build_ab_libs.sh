# Timing-only / debug builds of the CURRENT source for the A/B tools (tools/ab_roles.sh, tools/roles_span.py, tools/ab_train_fwd.py):
# tools/ab_libs/libadfp_<name>.so, selected at run time with ADFP_LIB_PATH.  Git-ignored; they travel with gpurun.
set -e
cd "$(dirname "$0")/../attentive_dfprior_amd/csrc"
OUT=../../tools/ab_libs
mkdir -p $OUT
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I../../include -shared -fPIC"
one() { /opt/rocm/bin/hipcc $FLAGS -o $OUT/libadfp_$1.so adfp_kernels.hip $2; echo built $1; }
one role0 -DADFP_EXP_ONLY_ROLE=0 &
one role1 -DADFP_EXP_ONLY_ROLE=1 &
one role2 -DADFP_EXP_ONLY_ROLE=2 &
one roles_span -DADFP_STAMPS_ROLES &
wait
one tune_shares -DADFP_TUNE_ROLE_SHARES &
one train_NOX -DADFP_EXP_TRAIN_NOX &
one train_NOC -DADFP_EXP_TRAIN_NOC &
one pb_mfma -DADFP_PB_MFMA &
wait
