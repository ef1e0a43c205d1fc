cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f4; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 1500 python tools/collect_roofline_counters.py > $O/counters.log 2>&1; cp gpurun_out/roofline_counters.json $O/; cp gpurun_out/roofline_counters.json profiles/roofline_counters.json
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json
for f in ['$O/bench.json']:
    d=[json.loads(l) for l in open(f) if l.startswith('{')][0]
    print(d['dtype'], d['value'], d['ms_per_step'], d.get('parity_rel_inf'))
    for k,v in d.items():
        if k.startswith('roofline') and v: print('  ',k, v['frac'], v.get('traffic'))"
