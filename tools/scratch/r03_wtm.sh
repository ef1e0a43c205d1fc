cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/wtm; mkdir -p $O
export FMC_AUTOTUNE_CACHE=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "tile_major or consumers_layernorm or consuming_gemm or statistics_from_the_producing or 160x320" > $O/pytest1.log 2>&1; tail -6 $O/pytest1.log
timeout 600 python bench.py --no-cpu-baseline --steps 20 > $O/bench_on.json 2> $O/bench_on.err; python -c "
import json;d=json.loads(open('$O/bench_on.json').read().strip().splitlines()[-1]);print('W tile-major ON :',d['ms_per_step'])"
FMC_W_TILEMAJOR=0 timeout 600 python bench.py --no-cpu-baseline --steps 20 > $O/bench_off.json 2> $O/bench_off.err; python -c "
import json;d=json.loads(open('$O/bench_off.json').read().strip().splitlines()[-1]);print('W tile-major OFF:',d['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --steps 20 > $O/bench_on2.json 2> $O/bench_on2.err; python -c "
import json;d=json.loads(open('$O/bench_on2.json').read().strip().splitlines()[-1]);print('W tile-major ON :',d['ms_per_step'])"
