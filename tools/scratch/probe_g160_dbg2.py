#!/usr/bin/env python
"""Is the LDS-DMA stream of gemm160_kernel bound per CU or chip-wide?  Same per-workgroup work on 256 / 64 / 16 / 8 workgroups."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from synfmc_amd import hip_ops as K
dev, bf = torch.device("cuda"), torch.bfloat16
f = (torch.randn(640, 640, 3, 3, device=dev, dtype=bf) * (9 * 640) ** -0.5).contiguous(memory_format=torch.channels_last)
out = []
for n in (32, 8, 2, 1):
    x = torch.randn(n, 20, 32, 640, device=dev, dtype=bf)
    t = K._time_ms(lambda: K.conv3x3_bf16(x, f, None, None, None, tile=512))
    out.append(f"{n * 8:4d} WGs: {t * 1e3:7.1f} us")
print(f"dbg={os.environ.get('FMC_G160_DBG', '0'):>2s}  conv 640->640 @20x32  " + "   ".join(out), flush=True)
