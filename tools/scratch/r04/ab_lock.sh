cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "8phase or stream_k or k_lockstep" > $O/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_k.log
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "cfg_shared or golden_g5 or pipeline" > $O/pytest_m.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_m.log
A="1,2,3,4,5,6,7,11,13,15,544,130,131,141,269"
FMC_AUTOTUNE_CACHE=$PWD/$O/at_a.json FMC_GEMM_ARMS=$A timeout 600 python bench.py --no-cpu-baseline > $O/bench_a.json 2> $O/bench_a.err; echo "A rc=$?"
FMC_AUTOTUNE_CACHE=$PWD/$O/at_b.json FMC_GEMM_ARMS=$A,397 timeout 600 python bench.py --no-cpu-baseline --autotune-log $O/autotune_b.log > $O/bench_b.json 2> $O/bench_b.err; echo "B rc=$?"
FMC_AUTOTUNE_CACHE=$PWD/$O/at_a.json FMC_GEMM_ARMS=$A timeout 600 python bench.py --no-cpu-baseline > $O/bench_a2.json 2> $O/bench_a2.err
FMC_AUTOTUNE_CACHE=$PWD/$O/at_b.json FMC_GEMM_ARMS=$A,397 timeout 600 python bench.py --no-cpu-baseline > $O/bench_b2.json 2> $O/bench_b2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04c/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['value'])
    except Exception as e: print(f, 'ERR', e)
PY
grep -c "arm 397" $O/autotune_b.log
