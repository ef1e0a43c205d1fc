cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/train; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "norm_with_skip" > $O/pytest1.log 2>&1; tail -8 $O/pytest1.log
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q > $O/pytest2.log 2>&1; tail -8 $O/pytest2.log
timeout 600 python bench.py --mode train --no-cpu-baseline > $O/train_on.json 2> $O/train_on.err; python -c "
import json;d=json.loads(open('$O/train_on.json').read().strip().splitlines()[-1]);print('norm-skip ON :',d['ms_per_step'])"
FMC_NORM_SKIP=0 timeout 600 python bench.py --mode train --no-cpu-baseline > $O/train_off.json 2> $O/train_off.err; python -c "
import json;d=json.loads(open('$O/train_off.json').read().strip().splitlines()[-1]);print('norm-skip OFF:',d['ms_per_step'])"
timeout 600 python bench.py --mode train --no-cpu-baseline > $O/train_on2.json 2> $O/train_on2.err; python -c "
import json;d=json.loads(open('$O/train_on2.json').read().strip().splitlines()[-1]);print('norm-skip ON :',d['ms_per_step'])"
