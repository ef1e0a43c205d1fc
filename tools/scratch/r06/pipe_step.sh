cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "geglu_ln_pipe or geglu_ln_direct" 2>&1 | tail -3
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
for v in 0 1 0 1; do
  FMC_GEGLU_PIPE=$v timeout 900 python bench.py --no-cpu-baseline --no-in-step --no-fp32-line --no-loop50 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pipe=$v', d['ms_per_step'])"
done
