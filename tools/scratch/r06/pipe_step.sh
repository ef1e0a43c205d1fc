cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "geglu_ln_pipe or geglu_ln_direct" 2>&1 | tail -3
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
for v in "0 0" "1 0" "1 1" "0 0" "1 0" "1 1"; do set -- $v
  FMC_GEGLU_PIPE=$1 FMC_GEGLU_PIPE_640=$2 timeout 900 python bench.py --no-cpu-baseline --no-in-step --no-fp32-line --no-loop50 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pipe=$1 pipe640=$2', d['ms_per_step'], d['parity_rel_inf'])"
done
