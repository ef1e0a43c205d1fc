cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04u; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
FMC_XATTN_FUSED_640=0 timeout 900 python bench.py --no-cpu-baseline > /dev/null 2>&1
FMC_XATTN_FUSED_640=1 timeout 900 python bench.py --no-cpu-baseline > /dev/null 2>&1
for v in 0 1 0 1 0 1; do
  FMC_XATTN_FUSED_640=$v timeout 900 python bench.py --no-cpu-baseline 2>$O/err_$v.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('xattn640=$v', d['value'], d['ms_per_step'])"
done
