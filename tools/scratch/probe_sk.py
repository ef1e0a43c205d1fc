#!/usr/bin/env python
"""Plain vs stream-K form of the 8-phase GEMM on one shape, a few launches each (for rocprofv3 traces / PMC passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from synfmc_amd import hip_ops as K
M, N, Kd = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 4096, 4096)))
a = torch.randn(M, Kd, device="cuda", dtype=torch.bfloat16)
w = torch.randn(N, Kd, device="cuda", dtype=torch.bfloat16) * 0.02
for t in (13, 141, 3, 131):
    for _ in range(5):
        K.linear_bf16(a, w, None, tile=t)
torch.cuda.synchronize()
