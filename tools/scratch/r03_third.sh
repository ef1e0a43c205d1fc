cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline --steps 20 > $O/bench_a.json 2> $O/bench_a.err; python -c "
import json;d=json.loads(open('$O/bench_a.json').read().strip().splitlines()[-1]);print('edge convs own :',d['ms_per_step'])"
FMC_PADDED_EDGE_CONVS=0 timeout 600 python bench.py --no-cpu-baseline --steps 20 > $O/bench_b.json 2> $O/bench_b.err; python -c "
import json;d=json.loads(open('$O/bench_b.json').read().strip().splitlines()[-1]);print('edge convs lib :',d['ms_per_step'])"
python - <<'P'
import json
d=json.load(open('gpurun_out/r03c/autotune_cache.json'))['choices']
for k,v in d.items():
    if "'conv'" in k and (", 64, 320," in k or ", 320, 8," in k): print(k, v['arm'], v['ms'])
P
