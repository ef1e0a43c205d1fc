#!/usr/bin/env python
"""Timing of the spatial self-attention backward (rowdot + dQ + dK/dV kernels) at the training shapes (bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from synfmc_amd import hip_ops as K
for (B, S, H, D) in [(16, 1536, 8, 40), (16, 384, 8, 80), (32, 4096, 8, 40)]:
    C = H * D
    qkv = torch.randn(B, S, 3 * C, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    out = K.spatial_attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], H)
    g = torch.randn_like(out)
    for _ in range(3):
        torch.autograd.grad(out, qkv, g, retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        torch.autograd.grad(out, qkv, g, retain_graph=True)
    e1.record()
    torch.cuda.synchronize()
    print(f"B={B} S={S} d={D}: backward {e0.elapsed_time(e1) / 10:8.3f} ms", flush=True)
