cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ln; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "consumers_layernorm or consuming_gemm" > $O/pytest1.log 2>&1; tail -12 $O/pytest1.log
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -s -k "bf16_vs_reference" > $O/pytest2.log 2>&1; grep -v amdgpu.ids $O/pytest2.log | tail -12
timeout 600 python bench.py --no-cpu-baseline --steps 20 > $O/bench_ln.json 2> $O/bench_ln.err; python -c "
import json;d=json.loads(open('$O/bench_ln.json').read().strip().splitlines()[-1]);print('LN epilogue ON :',d['ms_per_step'])"
FMC_LN_EPILOGUE=0 timeout 600 python bench.py --no-cpu-baseline --steps 20 > $O/bench_noln.json 2> $O/bench_noln.err; python -c "
import json;d=json.loads(open('$O/bench_noln.json').read().strip().splitlines()[-1]);print('LN epilogue OFF:',d['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --steps 20 > $O/bench_ln2.json 2> $O/bench_ln2.err; python -c "
import json;d=json.loads(open('$O/bench_ln2.json').read().strip().splitlines()[-1]);print('LN epilogue ON :',d['ms_per_step'])"
