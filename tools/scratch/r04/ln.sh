cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "ln or layernorm or 160 or temporal_block" > $O/pytest_ln.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_ln.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_model.py -q -m gpu -x > $O/pytest_model.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_model.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json
d=[json.loads(l) for l in open('$O/bench.json') if l.startswith('{')][0]
print(d['value'], d['ms_per_step'])"
