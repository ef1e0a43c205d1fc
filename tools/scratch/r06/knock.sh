# where does geglu_direct_kernel<320> spend its 204 us?  knock-out builds (bits: 1 = no weight loads, 2 = no LDS fragment reads, 4 = no epilogue)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/scratch/r05/bench_geglu.py 2>&1 | grep geglu | sed 's/^/full        /'
for n in 1 2 3 4 5 6 7; do
  FMC_HIP_LIB=$PWD/synfmc_amd/lib/knock/libfmc_hip_k$n.so python tools/scratch/r05/bench_geglu.py 2>&1 | grep geglu | sed "s/^/knock $n     /"
done
for n in 4 5 7; do
  FMC_GEGLU320_ROWS160=2 FMC_HIP_LIB=$PWD/synfmc_amd/lib/knock/libfmc_hip_k$n.so python tools/scratch/r05/bench_geglu.py 2>&1 | grep geglu320 | sed "s/^/SEQ knock $n /"
done
