"""Token projections on tile 16: persistent form (3-buffer ring, requests 2 sub-tiles ahead) vs the plain grid (5-buffer ring, 3 ahead).
Run twice: FMC_G160_PERSIST=1 / 0."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from synfmc_amd import hip_ops as K

torch.manual_seed(0)
print("FMC_G160_PERSIST =", os.environ.get("FMC_G160_PERSIST", "1"))
for (M, N, Kd, res) in [(81920, 320, 1280, True), (81920, 960, 320, False), (81920, 320, 320, True), (20480, 640, 2560, True), (20480, 1920, 640, False)]:
    x = torch.randn(M, Kd, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, Kd, device="cuda", dtype=torch.bfloat16) * Kd ** -0.5
    b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    r = torch.randn(M, N, device="cuda", dtype=torch.bfloat16) if res else None
    ms = K._time_ms(lambda: K.linear_bf16(x, w, b, r, 1.0, tile=512), reps=10)
    print(f"lin {M}x{N}x{Kd} res={res}: {ms * 1e3:7.1f} us ({2.0 * M * N * Kd / ms / 1e9:6.0f} TF/s)", flush=True)
