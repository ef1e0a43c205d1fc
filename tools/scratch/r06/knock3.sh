cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/scratch/r05/bench_geglu.py 2>&1 | grep geglu | sed 's/^/full      /'
for n in prio stag1 stag2 biaspf sb all1; do
  FMC_HIP_LIB=$PWD/synfmc_amd/lib/knock/libfmc_hip_$n.so python tools/scratch/r05/bench_geglu.py 2>&1 | grep geglu | sed "s/^/$n     /"
done
FMC_HIP_LIB=$PWD/synfmc_amd/lib/knock/libfmc_hip_stampall.so python tools/scratch/r06/stamp_geglu.py 2>&1 | tail -6
FMC_HIP_LIB=$PWD/synfmc_amd/lib/knock/libfmc_hip_all1.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "geglu_ln_direct" 2>&1 | tail -2
