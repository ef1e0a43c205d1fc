"""Fused temporal block at the 20x32 level (C = 640): time per launch, with / without the Camera-Adapter merge."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from synfmc_amd import hip_ops as K
B, Fr, hw, C, H = 2, 16, 640, 640, 8
dt = torch.bfloat16
h = torch.randn(B, Fr, hw, C, device="cuda", dtype=dt); pt = torch.randn_like(h)
g = torch.randn(C, device="cuda") * 0.2 + 1; bpe = torch.randn(Fr, C, device="cuda")
wq = torch.randn(3 * C, C, device="cuda", dtype=dt) * C ** -0.5
wo = torch.randn(C, C, device="cuda", dtype=dt) * C ** -0.5
wm = torch.randn(C, C, device="cuda", dtype=dt) * C ** -0.5
bo = torch.randn(C, device="cuda", dtype=dt)
wqp, wop, wmp = K.pack_temporal_qkv80(wq), K.pack_w_frag80(wo), K.pack_w_frag80(wm)
M = B * Fr * hw
for merge in (True, False):
    kw = dict(w_merge_tm=wmp, pose_term=pt, merge_scale=1.0) if merge else {}
    t = K._time_ms(lambda: K.temporal_block(h, g, bpe, 1e-5, wqp, wop, bo, 80 ** -0.5, **kw), reps=20)
    fl = 2.0 * M * C * (5 if merge else 4) * C + 4.0 * M * Fr * C
    print(f"fused block640 merge={merge}: {t*1e3:7.1f} us  {fl/t/1e9:6.0f} TF/s  frac {fl/t/1e9/2500:.3f}", flush=True)
