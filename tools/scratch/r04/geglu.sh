cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04z; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "geglu_ln_direct" > $O/pytest_g.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_g.log | cut -c1-300
timeout 120 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch
from synfmc_amd import hip_ops as K
from synfmc_amd.models.layers import interleave_geglu
M,C,cff=20480,640,2560
dt=torch.bfloat16
h=torch.randn(M,C,device="cuda",dtype=dt); g=torch.randn(C,device="cuda")*0.2+1; b=torch.randn(C,device="cuda")
w=torch.randn(2*cff,C,device="cuda",dtype=dt)*C**-0.5; bi=torch.randn(2*cff,device="cuda",dtype=dt)
wp=K.pack_geglu_frag80(w)
t=K._time_ms(lambda: K.geglu_ln_direct(h,g,b,1e-5,wp,bi,cff), reps=20)
print(f"geglu_ln_direct 20480x5120x640: {t*1e3:.1f} us  {2.0*M*2*cff*C/t/1e9:.0f} TF/s")
wi8,bi8=interleave_geglu(w,bi,8)
t2=K._time_ms(lambda: K.geglu_linear_blocked(K.layernorm(h,g,b,1e-5),wi8,bi8), reps=20)
print(f"layernorm + 160x320 GEGLU (tile-major out): {t2*1e3:.1f} us")
PY
