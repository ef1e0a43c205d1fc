#!/usr/bin/env python
"""Timing of the GroupNorm(+SiLU) backward (partial + apply kernels) at the training shapes (bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from synfmc_amd import hip_ops as K
for (N, H, W, C) in [(16, 32, 48, 320), (16, 16, 24, 640), (16, 8, 12, 1280), (16, 32, 48, 640)]:
    x = torch.randn(N, H * W, C, device="cuda", dtype=torch.bfloat16).requires_grad_(True)
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    y = K.groupnorm_silu(x, g, b, 32, 1e-5, True)
    dy = torch.randn_like(y)
    for _ in range(3):
        torch.autograd.grad(y, x, dy, retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        torch.autograd.grad(y, x, dy, retain_graph=True)
    e1.record()
    torch.cuda.synchronize()
    print(f"N={N} {H}x{W} C={C}: GN backward {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", flush=True)
