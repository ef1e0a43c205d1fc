cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "temporal_transformer_block_fused" > $O/pytest_a.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_a.log | cut -c1-250
timeout 1200 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu > $O/pytest_b.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_b.log | cut -c1-250
FMC_AUTOTUNE_CACHE=$PWD/$O/at.json timeout 600 python bench.py --no-cpu-baseline > $O/bench_fused.json 2> $O/bench_fused.err; echo "fused rc=$?"
FMC_TEMPORAL_FUSED=0 FMC_AUTOTUNE_CACHE=$PWD/$O/at.json timeout 600 python bench.py --no-cpu-baseline > $O/bench_plain.json 2> $O/bench_plain.err; echo "plain rc=$?"
FMC_AUTOTUNE_CACHE=$PWD/$O/at.json timeout 600 python bench.py --no-cpu-baseline > $O/bench_fused2.json 2> $O/bench_fused2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04e/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['value'], d.get('parity_rel_inf'))
    except Exception as e: print(f, 'ERR', e); print(open(f.replace('.json','.err')).read()[-1500:])
PY
