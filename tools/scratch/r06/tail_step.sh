# ff_tail (feed-forward output projection folded with proj_out, gemm160p_kernel's two-segment reduction): kernel test, isolated timing, step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "ff_tail or from_blocked or ffblk or blocked" 2>&1 | tail -5
timeout 600 python - <<'PY' 2>&1 | tail -8
import torch, bench
from synfmc_amd import hip_ops as K
import os
dev = torch.device("cuda:0")
for tail in ("1", "0"):
    K.FF_TAIL = tail == "1"
    r = bench.measure_ff2_roofline(dev, torch.bfloat16)
    print("FF_TAIL", tail, r["kernel"][-24:], "ms", r["avg_launch_ms"], "GB/s", r["achieved"], "mfma", r["mfma_frac_isolated"])
K.FF_TAIL = True
r = bench.measure_linear_l0_roofline(dev, torch.bfloat16)
print("linear_l0", r["avg_launch_ms"], r["achieved"])
PY
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
for v in 0 1 0 1; do
  FMC_FF_TAIL=$v timeout 900 python bench.py --no-cpu-baseline --no-in-step --no-fp32-line --no-loop50 2>$O/err_$v.log | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tail=$v', d['ms_per_step'], d['parity_rel_inf'])"
done
tail -3 $O/err_1.log
