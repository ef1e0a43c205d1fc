cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04y; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
: > $O/config_bench_lines.jsonl
for c in lora cam; do
  timeout 900 python bench.py --config $c --no-cpu-baseline 2>$O/err_$c.log | grep '^{' >> $O/config_bench_lines.jsonl
done
timeout 900 python bench.py --mode train --no-cpu-baseline 2>$O/err_train.log | grep '^{' >> $O/config_bench_lines.jsonl
timeout 1200 python bench.py --config train32 --fp8-temporal --no-cpu-baseline 2>$O/err_train32.log | grep '^{' >> $O/config_bench_lines.jsonl
python -c "
import json
for l in open('$O/config_bench_lines.jsonl'):
    d=json.loads(l); print(d['config'].get('baseline_config'), d['value'], d['ms_per_step'])"
