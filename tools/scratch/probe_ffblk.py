"""Feed-forward pair on tile 16: row-major vs tile-major intermediate (tools/r03: FMC_FF_BLOCKED)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from synfmc_amd import hip_ops as K
from synfmc_amd.models.layers import interleave_geglu

torch.manual_seed(0)
for (M, C) in [(81920, 320), (20480, 640)]:
    Cff = 4 * C
    x = torch.randn(M, C, device="cuda", dtype=torch.bfloat16)
    r = torch.randn(M, C, device="cuda", dtype=torch.bfloat16)
    w1 = torch.randn(2 * Cff, C, device="cuda", dtype=torch.bfloat16) * C ** -0.5
    b1 = torch.randn(2 * Cff, device="cuda", dtype=torch.bfloat16)
    w2 = torch.randn(C, Cff, device="cuda", dtype=torch.bfloat16) * Cff ** -0.5
    b2 = torch.randn(C, device="cuda", dtype=torch.bfloat16)
    wi8, bi8 = interleave_geglu(w1, b1, 8)
    mid_rm = K.linear_bf16(x, wi8, bi8, geglu=True, tile=512)
    mid_tm = K.geglu_linear_blocked(x, wi8, bi8)
    t = {}
    t["geglu row-major"] = K._time_ms(lambda: K.linear_bf16(x, wi8, bi8, geglu=True, tile=512), reps=10)
    t["geglu tile-major"] = K._time_ms(lambda: K.geglu_linear_blocked(x, wi8, bi8), reps=10)
    t["ff-out row-major"] = K._time_ms(lambda: K.linear_bf16(mid_rm, w2, b2, r, 1.0, tile=512), reps=10)
    t["ff-out tile-major"] = K._time_ms(lambda: K.linear_from_blocked(mid_tm, w2, b2, r), reps=10)
    t["pair row-major"] = K._time_ms(lambda: K.linear_bf16(K.linear_bf16(x, wi8, bi8, geglu=True, tile=512), w2, b2, r, 1.0, tile=512), reps=10)
    t["pair tile-major"] = K._time_ms(lambda: K.linear_from_blocked(K.geglu_linear_blocked(x, wi8, bi8), w2, b2, r), reps=10)
    print(f"M={M} C={C}: " + "  ".join(f"{k} {v * 1e3:.1f} us" for k, v in t.items()), flush=True)
