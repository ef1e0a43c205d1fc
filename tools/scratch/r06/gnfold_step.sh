# GroupNorm folded into proj_in (per-image weights): step A/B, then the model-level parity tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
for v in 0 1 0 1; do
  FMC_GN_FOLD=$v timeout 900 python bench.py --no-cpu-baseline --no-in-step --no-fp32-line --no-loop50 2>$O/err_$v.log | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('gn_fold=$v', d['ms_per_step'], d['step_dispatch']['linear'])"
done
tail -2 $O/err_1.log
timeout 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_model.py tests/test_gpu_full_width.py -q -m gpu -x 2>&1 | tail -5
