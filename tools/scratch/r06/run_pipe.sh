cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "geglu_ln_pipe" 2>&1 | tail -2
VAR=1 python tools/scratch/r06/bench_pipe.py 2>&1 | grep "   direct " | head -1
for n in vpm3 vpm4; do echo $n; VAR=1 FMC_HIP_LIB=$PWD/synfmc_amd/lib/knock/libfmc_hip_$n.so python tools/scratch/r06/bench_pipe.py 2>&1 | grep "   direct " | head -1; done
