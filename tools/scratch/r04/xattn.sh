cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04t; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "xattn_block_fused" > $O/pytest_x.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_x.log | cut -c1-400
timeout 120 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch
from synfmc_amd import hip_ops as K
B,Fr,hw,C,S=2,16,640,640,77
dt=torch.bfloat16
h=torch.randn(B*Fr,hw,C,device="cuda",dtype=dt); g=torch.randn(C,device="cuda")*0.2+1; bt=torch.randn(16,C,device="cuda")
wq=K.pack_w_frag80(torch.randn(C,C,device="cuda",dtype=dt)*C**-0.5); wo=K.pack_w_frag80(torch.randn(C,C,device="cuda",dtype=dt)*C**-0.5)
bo=torch.randn(C,device="cuda",dtype=dt); kv=torch.randn(B,S,2*C,device="cuda",dtype=dt)
t=K._time_ms(lambda: K.xattn_block640(h,g,bt,1e-5,wq,kv,wo,bo,80**-0.5,Fr), reps=20)
print(f"xattn block640 (pack + block): {t*1e3:.1f} us")
B,Fr,hw,C,S=2,16,2560,320,77
h=torch.randn(B*Fr,hw,C,device="cuda",dtype=dt); g=torch.randn(C,device="cuda")*0.2+1; bt=torch.randn(16,C,device="cuda")
wq=K.pack_xattn_q40(torch.randn(C,C,device="cuda",dtype=dt)*C**-0.5); wo=K._w_tilemajor(torch.randn(C,C,device="cuda",dtype=dt)*C**-0.5)
bo=torch.randn(C,device="cuda",dtype=dt); kv=torch.randn(B,S,2*C,device="cuda",dtype=dt)
t=K._time_ms(lambda: K.xattn_block(h,g,bt,1e-5,wq,kv,wo,bo,40**-0.5,Fr,stats_eps=1e-5), reps=20)
print(f"xattn block320 (pack + block + stats): {t*1e3:.1f} us")
PY
