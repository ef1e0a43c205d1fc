#!/usr/bin/env python
"""Graph-timed TFLOP/s of the fmc GEMM/conv arms on a few FMC shapes (GPU box only).  usage: probe_gemm2.py [tiles...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from synfmc_amd import hip_ops as K
from synfmc_amd.models.layers import interleave_geglu
dev, dt = "cuda", torch.bfloat16
tiles = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 5, 7]


def row(name, fl, fns):
    print(f"{name:34s}" + " ".join(f"{k}:{fl / K._time_ms(f) / 1e9:6.0f}" for k, f in fns), flush=True)


for n, ci, co, h, w_ in [(32, 1280, 1280, 20, 32), (32, 320, 320, 40, 64), (32, 640, 640, 20, 32), (32, 1280, 1280, 10, 16)]:
    x = torch.randn(n, h, w_, ci, device=dev, dtype=dt)
    wt = (torch.randn(co, ci, 3, 3, device=dev, dtype=dt) * 0.02).contiguous(memory_format=torch.channels_last)
    xn = x.permute(0, 3, 1, 2)
    row(f"conv {n}x{h}x{w_} {ci}->{co}", 2.0 * n * h * w_ * ci * co * 9,
        [("lib", lambda: F.conv2d(xn, wt, None, 1, 1))] + [(t, lambda t=t: K.conv3x3_bf16(x, wt, None, None, None, tile=t)) for t in tiles])
for M, Kd, N in [(20480, 2560, 2560), (81920, 320, 320), (81920, 320, 960), (20480, 640, 640), (5120, 1280, 10240), (5120, 1280, 1280)]:
    a = torch.randn(M, Kd, device=dev, dtype=dt)
    w2 = torch.randn(N, Kd, device=dev, dtype=dt) * 0.02
    row(f"lin M={M} K={Kd} N={N}", 2.0 * M * Kd * N,
        [("lib", lambda: F.linear(a, w2))] + [(t, lambda t=t: K.linear_bf16(a, w2, None, tile=t)) for t in tiles])
for M, C in [(81920, 320), (20480, 640)]:
    a = torch.randn(M, C, device=dev, dtype=dt)
    w = torch.randn(8 * C, C, device=dev, dtype=dt) * 0.02
    wi, _ = interleave_geglu(w, None)
    row(f"geglu M={M} C={C}", 2.0 * M * C * 8 * C,
        [("lib", lambda: K.geglu(F.linear(a, w)))] + [(t, lambda t=t: K.linear_bf16(a, wi, None, geglu=True, tile=t)) for t in tiles])
print("--- with residual epilogue (us per call)")
for M, Kd, N in [(81920, 320, 320), (20480, 640, 640), (5120, 1280, 1280), (81920, 1280, 320)]:
    a = torch.randn(M, Kd, device=dev, dtype=dt)
    w2 = torch.randn(N, Kd, device=dev, dtype=dt) * 0.02
    b2 = torch.zeros(N, device=dev, dtype=dt)
    r2 = torch.randn(M, N, device=dev, dtype=dt)
    print(f"lin+res M={M} K={Kd} N={N}: lib {K._time_ms(lambda: F.linear(a, w2, b2) + r2) * 1e3:6.1f} | " +
          " ".join(f"{t}:{K._time_ms(lambda t=t: K.linear_bf16(a, w2, b2, r2, tile=t)) * 1e3:6.1f}" for t in tiles), flush=True)
for n, ci, co, h, w_ in [(32, 320, 320, 40, 64), (32, 1280, 1280, 10, 16)]:
    x = torch.randn(n, h, w_, ci, device=dev, dtype=dt)
    wt = (torch.randn(co, ci, 3, 3, device=dev, dtype=dt) * 0.02).contiguous(memory_format=torch.channels_last)
    r = torch.randn(n, h, w_, co, device=dev, dtype=dt)
    print(f"conv+res {n}x{h}x{w_} {ci}->{co}: " +
          " ".join(f"{t}:{K._time_ms(lambda t=t: K.conv3x3_bf16(x, wt, None, None, r, tile=t)) * 1e3:6.1f}" for t in tiles), flush=True)
