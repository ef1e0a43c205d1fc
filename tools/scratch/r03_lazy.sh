cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/lazy; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "residual_add_in_front" > $O/pytest1.log 2>&1; tail -5 $O/pytest1.log
timeout 1200 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_full_width.py -x -q > $O/pytest2.log 2>&1; tail -5 $O/pytest2.log
timeout 600 python bench.py --no-cpu-baseline --steps 20 > $O/bench_on.json 2> $O/bench_on.err; python -c "
import json;d=json.loads(open('$O/bench_on.json').read().strip().splitlines()[-1]);print('lazy residual ON :',d['ms_per_step'], d.get('roofline_proj'))"
FMC_LAZY_RESIDUAL=0 timeout 600 python bench.py --no-cpu-baseline --steps 20 > $O/bench_off.json 2> $O/bench_off.err; python -c "
import json;d=json.loads(open('$O/bench_off.json').read().strip().splitlines()[-1]);print('lazy residual OFF:',d['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --steps 20 > $O/bench_on2.json 2> $O/bench_on2.err; python -c "
import json;d=json.loads(open('$O/bench_on2.json').read().strip().splitlines()[-1]);print('lazy residual ON :',d['ms_per_step'])"
