cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04x2; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "xattn_block_fused_1280" > $O/pytest_x.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_x.log | cut -c1-300
timeout 120 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch
from synfmc_amd import hip_ops as K
B,Fr,hw,C,S=2,16,160,1280,77
dt=torch.bfloat16
h=torch.randn(B*Fr,hw,C,device="cuda",dtype=dt); g=torch.randn(C,device="cuda")*0.2+1; bt=torch.randn(C,device="cuda")
wq=K.pack_w_frag160(torch.randn(C,C,device="cuda",dtype=dt)*C**-0.5); wo=K.pack_w_frag160(torch.randn(C,C,device="cuda",dtype=dt)*C**-0.5)
bo=torch.randn(C,device="cuda",dtype=dt); kv=torch.randn(B,S,2*C,device="cuda",dtype=dt)
t=K._time_ms(lambda: K.xattn_block(h,g,bt,1e-5,wq,kv,wo,bo,160**-0.5,Fr), reps=20)
print(f"xattn block1280 (pack + block): {t*1e3:.1f} us")
PY
