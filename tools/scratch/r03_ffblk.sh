cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ffblk; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "tile_major" > $O/pytest1.log 2>&1; tail -12 $O/pytest1.log
timeout 300 python tools/scratch/probe_ffblk.py 2>&1 | grep -v amdgpu | tee $O/probe.txt
timeout 600 python bench.py --no-cpu-baseline --steps 20 > $O/bench_on.json 2> $O/bench_on.err; python -c "
import json;d=json.loads(open('$O/bench_on.json').read().strip().splitlines()[-1]);print('ff blocked ON :',d['ms_per_step'], d['parity_rel_inf'])"
FMC_FF_BLOCKED=0 timeout 600 python bench.py --no-cpu-baseline --steps 20 > $O/bench_off.json 2> $O/bench_off.err; python -c "
import json;d=json.loads(open('$O/bench_off.json').read().strip().splitlines()[-1]);print('ff blocked OFF:',d['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --steps 20 > $O/bench_on2.json 2> $O/bench_on2.err; python -c "
import json;d=json.loads(open('$O/bench_on2.json').read().strip().splitlines()[-1]);print('ff blocked ON :',d['ms_per_step'])"
