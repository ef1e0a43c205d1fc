cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
python tools/scratch/r06/bench_pipe.py 2>&1 | grep "   direct "
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
for v in 0 1 0 1; do
  FMC_GEGLU_PIPE_640=$v timeout 900 python bench.py --no-cpu-baseline --no-in-step --no-fp32-line --no-loop50 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pipe640=$v', d['ms_per_step'])"
done
