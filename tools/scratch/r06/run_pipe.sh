cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
VAR=1 python tools/scratch/r06/bench_pipe.py 2>&1 | grep "   direct " | head -1
for n in afd3 afd4 ng1 ngafd4; do echo $n; VAR=1 FMC_HIP_LIB=$PWD/synfmc_amd/lib/knock/libfmc_hip_$n.so python tools/scratch/r06/bench_pipe.py 2>&1 | grep "   direct " | head -1; done
