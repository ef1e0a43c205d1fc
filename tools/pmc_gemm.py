#!/usr/bin/env python
"""One conv / GEMM shape, a few launches: the target of `rocprofv3 --pmc ...` counter passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synfmc_amd import hip_ops as K
dev, dt = "cuda", torch.bfloat16
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n, ci, co, h, w_ = 32, 1280, 1280, 20, 32
x = torch.randn(n, h, w_, ci, device=dev, dtype=dt)
wt = (torch.randn(co, ci, 3, 3, device=dev, dtype=dt) * 0.02).contiguous(memory_format=torch.channels_last)
for _ in range(3):
    K.conv3x3_bf16(x, wt, None, None, None, tile=tile)
M, Kd, N = 20480, 2560, 2560
a = torch.randn(M, Kd, device=dev, dtype=dt)
w2 = torch.randn(N, Kd, device=dev, dtype=dt) * 0.02
for _ in range(3):
    K.linear_bf16(a, w2, None, tile=tile)
torch.cuda.synchronize()
