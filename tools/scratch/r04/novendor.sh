cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04o; mkdir -p $O
for v in 0 1 0 1; do
  export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache_$v.json
  FMC_NO_VENDOR=$v timeout 900 python bench.py --no-cpu-baseline 2>$O/err_$v.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('no_vendor=$v', d['value'], d['ms_per_step'])"
done
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache_1.json
FMC_NO_VENDOR=1 timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $T > $O/kernel_summary.md 2>&1
find $O/trace -name "*.csv" -size +1M -delete
head -22 $O/kernel_summary.md | cut -c1-150; grep -c Cijk $O/kernel_summary.md
