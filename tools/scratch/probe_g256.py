"""GEGLU projections of the three U-Net levels: arm 528 (256 x 320 persistent) against 512 (160 x 320), 3 (256 x 256 ring), 13 (8-phase)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from synfmc_amd import hip_ops as K
from synfmc_amd.models.layers import interleave_geglu

torch.manual_seed(0)
for (M, N, Kd) in [(81920, 2560, 320), (40960, 2560, 320), (20480, 5120, 640), (10240, 5120, 640), (5120, 10240, 1280)]:
    x = torch.randn(M, Kd, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, Kd, device="cuda", dtype=torch.bfloat16) * Kd ** -0.5
    b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    w8, b8 = interleave_geglu(w, b, 8)
    w32, b32 = interleave_geglu(w, b)
    line = f"geglu {M}x{N}x{Kd}:"
    for arm in (K.ARM_256, K.ARM_160, 3, 13):
        wi, bi = (w8, b8) if arm >= 512 else (w32, b32)
        ms = K._time_ms(lambda: K.linear_bf16(x, wi, bi, geglu=True, tile=arm), reps=10)
        line += f"  arm {arm}: {ms * 1e3:7.1f} us ({2.0 * M * N * Kd / ms / 1e9:6.0f} TF/s)"
    print(line, flush=True)
