#!/usr/bin/env python
"""Timing of the temporal attention launches (bf16 forward, fp8 forward, bf16 backward) at the U-Net's level-0 / level-1 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from synfmc_amd import hip_ops as K


def bench(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (B, Fr, P, H, D) in [(2, 16, 2560, 8, 40), (2, 16, 640, 8, 80), (1, 32, 4096, 8, 40)]:
    C = H * D
    qkv = torch.randn(B, Fr, P, 3 * C, device="cuda", dtype=torch.bfloat16)
    q8 = torch.randn(B, Fr, P, 3 * C, device="cuda").clamp(-3, 3).to(torch.float8_e4m3fn)
    sc = torch.ones(3, device="cuda")
    mb = 4.0 * B * Fr * P * C * 2 / 1e6
    t = bench(lambda: K.self_attention_qkv(qkv, H, D ** -0.5, True))
    print(f"B={B} F={Fr} P={P} d={D}: bf16 fwd {t:7.1f} us ({mb / t:5.2f} TB/s algorithmic)", end="  ")
    t8 = bench(lambda: K._temporal_fp8_raw(q8, sc, H, D ** -0.5))
    print(f"fp8 fwd {t8:7.1f} us", end="  ")
    q = qkv.detach().clone().requires_grad_(True)
    out = K.self_attention_qkv(q, H, D ** -0.5, True)
    g = torch.randn_like(out)
    tb = bench(lambda: torch.autograd.grad(out, q, g, retain_graph=True))
    print(f"bf16 bwd (autograd node) {tb:7.1f} us", flush=True)
