cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04n; mkdir -p $O
for v in 1 2 1 2; do
  export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache_$v.json
  FMC_G160_PERSIST=$v timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('persist=$v', d['value'], d['ms_per_step'])"
done
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache_1.json
timeout 1500 python tools/collect_roofline_counters.py > $O/counters.log 2>&1; tail -3 $O/counters.log; cp gpurun_out/roofline_counters.json $O/
