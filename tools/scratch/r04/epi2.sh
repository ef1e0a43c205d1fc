cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
for v in old new old new; do
  if [ $v = new ]; then unset FMC_HIP_LIB; else export FMC_HIP_LIB=$PWD/tools/scratch/r04/libs/libfmc_$v.so; fi
  echo "== $v"; timeout 300 python tools/scratch/probe_ffblk.py 2>&1 | grep "^M=" | cut -c1-330
done
unset FMC_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "geglu or gemm or linear" > $O/pytest_geglu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_geglu.log | cut -c1-300
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
for v in old new old new; do
  if [ $v = new ]; then unset FMC_HIP_LIB; else export FMC_HIP_LIB=$PWD/tools/scratch/r04/libs/libfmc_$v.so; fi
  timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['value'], d['ms_per_step'])"
done
