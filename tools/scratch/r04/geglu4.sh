cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04z4; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "geglu_ln_direct" 2>&1 | tail -2
cat > /tmp/pg.py <<'PY'
import torch, os
from synfmc_amd import hip_ops as K
for (M,C,cff) in [(20480,640,2560),(81920,320,1280)]:
    dt=torch.bfloat16
    h=torch.randn(M,C,device="cuda",dtype=dt); g=torch.randn(C,device="cuda")*0.2+1; b=torch.randn(C,device="cuda")
    w=torch.randn(2*cff,C,device="cuda",dtype=dt)*C**-0.5; bi=torch.randn(2*cff,device="cuda",dtype=dt)
    wp=K.pack_geglu_frag80(w)
    t=K._time_ms(lambda: K.geglu_ln_direct(h,g,b,1e-5,wp,bi,cff), reps=20)
    print(os.environ.get("FMC_HIP_LIB","new")[-12:], f"geglu_ln_direct C={C}: {t*1e3:.1f} us")
PY
for v in old new old new; do
  if [ $v = new ]; then unset FMC_HIP_LIB; else export FMC_HIP_LIB=$PWD/tools/scratch/r04/libs/libfmc_old.so; fi
  PYTHONPATH=$PWD timeout 120 python /tmp/pg.py 2>&1 | grep geglu
done
unset FMC_HIP_LIB
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 900 python bench.py --no-cpu-baseline > /dev/null 2>&1
for v in old new old new old new; do
  if [ $v = new ]; then unset FMC_HIP_LIB; else export FMC_HIP_LIB=$PWD/tools/scratch/r04/libs/libfmc_old.so; fi
  timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['value'], d['ms_per_step'])"
done
