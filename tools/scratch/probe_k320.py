#!/usr/bin/env python
"""Where do the K = 320 projections of level 0 spend their time?  Same M, N with K = 64 (epilogue + launch only) vs
K = 320, per tile arm, with and without the residual; plus a pure copy of the same bytes for reference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from synfmc_amd import hip_ops as K
dev, dt = "cuda", torch.bfloat16


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(8):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters / 8 * 1e3


M = 81920
for N in (320, 960):
    res = torch.randn(M, N, device=dev, dtype=dt)
    b = torch.randn(N, device=dev, dtype=dt)
    for Kd in (64, 320):
        a = torch.randn(M, Kd, device=dev, dtype=dt)
        w = torch.randn(N, Kd, device=dev, dtype=dt) * 0.05
        for tile in (4, 5, 1, 2, 6, 11):
            t_r = timed(lambda: K.linear_bf16(a, w, b, res, 1.0, tile=tile))
            t_n = timed(lambda: K.linear_bf16(a, w, b, None, 1.0, tile=tile))
            print(f"N={N} K={Kd:4d} tile={tile:2d}: +res {t_r:6.1f} us   no res {t_n:6.1f} us", flush=True)
    src = torch.randn(M, N, device=dev, dtype=dt)
    dst = torch.empty_like(src)
    print(f"N={N}: copy [M,N] {timed(lambda: dst.copy_(src)):6.1f} us   add {timed(lambda: torch.add(src, res, out=dst)):6.1f} us", flush=True)

from synfmc_amd.models.layers import interleave_geglu
for (M, Kd, N) in [(81920, 320, 2560), (20480, 640, 5120), (5120, 1280, 10240)]:
    a = torch.randn(M, Kd, device=dev, dtype=dt)
    w = torch.randn(N, Kd, device=dev, dtype=dt) * 0.05
    wi, bi = interleave_geglu(w, torch.randn(N, device=dev, dtype=dt))
    for tile in (4, 5, 1, 2, 6):
        print(f"geglu M={M} K={Kd} N={N} tile={tile:3d}: {timed(lambda: K.linear_bf16(a, wi, bi, geglu=True, tile=tile)):7.1f} us", flush=True)
for (n, h, w_, ci, co) in [(32, 40, 64, 320, 320), (32, 20, 32, 640, 640)]:
    x = torch.randn(n, h, w_, ci, device=dev, dtype=dt)
    wt = (torch.randn(co, ci, 3, 3, device=dev, dtype=dt) * 0.02).contiguous(memory_format=torch.channels_last)
    for tile in (4, 5, 1, 2, 6, 11, 3):
        print(f"conv {n}x{h}x{w_} {ci}->{co} tile={tile:3d}: {timed(lambda: K.conv3x3_bf16(x, wt, None, None, None, tile=tile)):7.1f} us", flush=True)
