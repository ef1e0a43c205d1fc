cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04x2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_full_size.py -q -m gpu -x -k "fused_text_cross or bench_step or g5" > $O/pytest_model.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_model.log | cut -c1-300
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
FMC_XATTN_FUSED_1280=0 timeout 900 python bench.py --no-cpu-baseline > /dev/null 2>&1
FMC_XATTN_FUSED_1280=1 timeout 900 python bench.py --no-cpu-baseline > /dev/null 2>&1
for v in 0 1 0 1 0 1; do
  FMC_XATTN_FUSED_1280=$v timeout 900 python bench.py --no-cpu-baseline 2>$O/err_$v.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('x1280=$v', d['value'], d['ms_per_step'])"
done
