#!/usr/bin/env python
"""GEMM / conv shapes of the U-Net at fixed tile arms: timing (events) for the tile-order A/B (FMC_GEMM_GM) and the
target of the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_tile_order.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from synfmc_amd import hip_ops as K
dev, dt = "cuda", torch.bfloat16
iters = int(os.environ.get("PROBE_ITERS", "10"))


def timed(fn):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


# (label, M, K, N, epilogue-geglu, tile)
LIN = [("geglu L0", 81920, 320, 2560, True, 4), ("geglu L1", 20480, 640, 5120, True, 4), ("geglu L2", 5120, 1280, 10240, True, 4),
       ("ffout L1", 20480, 2560, 640, False, 3), ("ffout L2", 5120, 5120, 1280, False, 2), ("qkv L1", 20480, 640, 1920, False, 3),
       ("qkv L2", 5120, 1280, 3840, False, 3)]
for label, M, Kd, N, geglu, tile in LIN:
    a = torch.randn(M, Kd, device=dev, dtype=dt)
    w = torch.randn(N, Kd, device=dev, dtype=dt) * 0.02
    b = torch.randn(N, device=dev, dtype=dt)
    ms = timed(lambda: K.linear_bf16(a, w, b, None, 1.0, geglu=geglu, tile=tile) if geglu else K.linear_bf16(a, w, b, None, 1.0, tile=tile))
    fl = 2.0 * M * Kd * N
    alg = (M * Kd + N * Kd + M * (N // 2 if geglu else N)) * 2
    print(f"{label:9s} M={M} K={Kd} N={N} tile={tile}: {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s  algorithmic {alg / 1e6:.1f} MB", flush=True)
# (label, n, h, w, ci, co, tile)
CONV = [("conv L0", 32, 40, 64, 320, 320, 11), ("conv L1", 32, 20, 32, 640, 640, 3), ("conv L2", 32, 10, 16, 1280, 1280, 128 + 3),
        ("conv L1u", 32, 20, 32, 1280, 640, 3)]
for label, n, h, w_, ci, co, tile in CONV:
    x = torch.randn(n, h, w_, ci, device=dev, dtype=dt)
    wt = (torch.randn(co, ci, 3, 3, device=dev, dtype=dt) * 0.02).contiguous(memory_format=torch.channels_last)
    ms = timed(lambda: K.conv3x3_bf16(x, wt, None, None, None, tile=tile))
    fl = 2.0 * n * h * w_ * 9 * ci * co
    alg = (n * h * w_ * (ci + co) + 9 * ci * co) * 2
    print(f"{label:9s} {n}x{h}x{w_} {ci}->{co} tile={tile}: {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s  algorithmic {alg / 1e6:.1f} MB", flush=True)
