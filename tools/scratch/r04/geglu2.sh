cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_full_size.py -q -m gpu -x -k "geglu_ln_direct or fused_text_cross or fused_kernel_equals or bench_step or g5" > $O/pytest_model.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_model.log | cut -c1-300
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
FMC_GEGLU_DIRECT_640=0 timeout 900 python bench.py --no-cpu-baseline > /dev/null 2>&1
FMC_GEGLU_DIRECT_640=1 timeout 900 python bench.py --no-cpu-baseline > /dev/null 2>&1
for v in 0 1 0 1 0 1; do
  FMC_GEGLU_DIRECT_640=$v timeout 900 python bench.py --no-cpu-baseline 2>$O/err_$v.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('geglu_direct=$v', d['value'], d['ms_per_step'])"
done
