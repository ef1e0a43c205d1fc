cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04z2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_full_size.py -q -m gpu -x -k "geglu_ln_direct or fused_text_cross or fused_kernel_equals or bench_step or g5 or cfg_shared" > $O/pytest_model.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_model.log | cut -c1-300
timeout 120 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch
from synfmc_amd import hip_ops as K
M,C,cff=81920,320,1280
dt=torch.bfloat16
h=torch.randn(M,C,device="cuda",dtype=dt); g=torch.randn(C,device="cuda")*0.2+1; b=torch.randn(C,device="cuda")
w=torch.randn(2*cff,C,device="cuda",dtype=dt)*C**-0.5; bi=torch.randn(2*cff,device="cuda",dtype=dt)
wp=K.pack_geglu_frag80(w)
t=K._time_ms(lambda: K.geglu_ln_direct(h,g,b,1e-5,wp,bi,cff), reps=20)
print(f"geglu_ln_direct 81920x2560x320: {t*1e3:.1f} us  {2.0*M*2*cff*C/t/1e9:.0f} TF/s")
PY
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
FMC_GEGLU_DIRECT_320=0 timeout 900 python bench.py --no-cpu-baseline > /dev/null 2>&1
FMC_GEGLU_DIRECT_320=1 timeout 900 python bench.py --no-cpu-baseline > /dev/null 2>&1
for v in 0 1 0 1 0 1; do
  FMC_GEGLU_DIRECT_320=$v timeout 900 python bench.py --no-cpu-baseline 2>$O/err_$v.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('geglu320=$v', d['value'], d['ms_per_step'])"
done
