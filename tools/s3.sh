cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
bash tools/ab_fourier_mfma.sh > $O/s3_ab_fourier_mfma.txt 2>&1
cat $O/s3_ab_fourier_mfma.txt
