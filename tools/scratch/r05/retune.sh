# fresh per-shape arm selection (tracked defaults ignored) for the inference configs + training shapes -> caches to merge into the default table
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
export FMC_AUTOTUNE_DEFAULTS=0
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 900 python bench.py --no-cpu-baseline --no-in-step > $O/bench_obj.json 2> $O/bench_obj.err; tail -c 300 $O/bench_obj.err
for c in lora cam; do timeout 900 python bench.py --config $c --no-cpu-baseline --no-in-step > $O/bench_$c.json 2> $O/err_$c.log; done
timeout 900 python bench.py --mode train --no-cpu-baseline > $O/bench_train.json 2> $O/err_train.log
timeout 1200 python bench.py --config train32 --fp8-temporal --no-cpu-baseline > $O/bench_train32.json 2> $O/err_train32.log
python -c "
import json
for n in ('obj','lora','cam','train','train32'):
    try:
        d=json.load(open('$O/bench_%s.json' % n)); print(n, d['value'], d['ms_per_step'])
    except Exception as e: print(n, 'failed', e)
d=json.load(open('$O/autotune_cache.json')); import collections; print(len(d['choices']), collections.Counter(v['arm'] for v in d['choices'].values()).most_common(8))"
