cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
for v in 0 1 0 1; do
  FMC_PREFER_K320=$v timeout 900 python bench.py --no-cpu-baseline --no-in-step --no-fp32-line --no-loop50 2>$O/err_$v.log | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('prefer_k320=$v', d['ms_per_step'])"
done
