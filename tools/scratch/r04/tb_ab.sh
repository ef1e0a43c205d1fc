cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
for i in 1 2; do
FMC_HIP_LIB=$PWD/tools/scratch/r04/libs/libfmc_oldtb.so timeout 120 python tools/scratch/r04/probe_tb.py 2>&1 | grep "fused block" | sed 's/^/old /'
timeout 120 python tools/scratch/r04/probe_tb.py 2>&1 | grep "fused block" | sed 's/^/new /'
done
