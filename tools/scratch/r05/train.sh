cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 900 python bench.py --mode train --no-cpu-baseline 2>$O/err_train.log | grep '^{' > $O/train_lines.jsonl
timeout 1200 python bench.py --config train32 --fp8-temporal --no-cpu-baseline 2>$O/err_train32.log | grep '^{' >> $O/train_lines.jsonl
python -c "
import json
for l in open('$O/train_lines.jsonl'):
    d=json.loads(l); print(d['config'].get('workload','')[:60], d['value'], d['ms_per_step'])"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_train -o t --output-format csv -- python bench.py --mode train --no-cpu-baseline --steps 5 --warmup 2 > $O/trace_train.log 2>&1
T=$(find $O/trace_train -name "*kernel_trace.csv" | head -1)
python tools/summarize_train_trace.py $T > $O/train_kernel_summary.md 2>&1
find $O/trace_train -name "*.csv" -size +1M -delete
head -30 $O/train_kernel_summary.md | cut -c1-170
