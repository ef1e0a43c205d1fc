cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "small_m or 8phase_arms or gemm_arms or linear" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-300
for v in 0 1 0 1; do
  export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache_$v.json
  FMC_NO_VENDOR=$v timeout 900 python bench.py --no-cpu-baseline 2>$O/err_$v.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('no_vendor=$v', d['value'], d['ms_per_step'])"
done
