"""LayerNorm launches of levels 1-3: rows per lane group (FMC_LN_U)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from synfmc_amd import hip_ops as K
for (M, C) in [(5120, 1280), (1280, 1280), (2560, 1280), (20480, 640), (10240, 640), (81920, 320)]:
    x = torch.randn(M, C, device="cuda", dtype=torch.bfloat16)
    g = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
    t = K._time_ms(lambda: K.layernorm(x, g, b), reps=30)
    print(f"U={os.environ.get('FMC_LN_U','-')} M={M} C={C}: {t*1e3:.1f} us  {2*M*C*2/t/1e9:.2f} TB/s", flush=True)
