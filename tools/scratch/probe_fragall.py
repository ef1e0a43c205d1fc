#!/usr/bin/env python
"""Deep-K GEMM / conv shapes on the <= 8-wave tile arms (whole-k-tile fragment preload) next to the 16-wave 256x256 arm.
A/B across builds: FMC_HIP_LIB=<other libfmc_hip.so>."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from synfmc_amd import hip_ops as K
dev, dt = "cuda", torch.bfloat16
for (M, Kd, N) in [(5120, 5120, 1280), (5120, 1280, 3840), (20480, 2560, 640), (20480, 640, 1920)]:
    a = torch.randn(M, Kd, device=dev, dtype=dt)
    w = torch.randn(N, Kd, device=dev, dtype=dt) * 0.02
    for tile in (1, 2, 7, 3):
        ms = K._time_ms(lambda: K.linear_bf16(a, w, None, None, 1.0, tile=tile))
        print(f"lin M={M} K={Kd} N={N} tile={tile}: {ms * 1e3:7.1f} us {2.0 * M * Kd * N / ms / 1e9:7.1f} TF/s", flush=True)
for (n, h, w_, ci, co) in [(32, 20, 32, 640, 640), (32, 10, 16, 1280, 1280)]:
    x = torch.randn(n, h, w_, ci, device=dev, dtype=dt)
    wt = (torch.randn(co, ci, 3, 3, device=dev, dtype=dt) * 0.02).contiguous(memory_format=torch.channels_last)
    for tile in (1, 2, 7, 3):
        ms = K._time_ms(lambda: K.conv3x3_bf16(x, wt, None, None, None, tile=tile))
        print(f"conv {n}x{h}x{w_} {ci}->{co} tile={tile}: {ms * 1e3:7.1f} us {2.0 * n * h * w_ * 9 * ci * co / ms / 1e9:7.1f} TF/s", flush=True)
