#!/usr/bin/env python
"""fmc_linear_bf16 / fmc_conv3x3_bf16 vs hipBLASLt / MIOpen at the 16x320x512 CFG-2 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
import synfmc_amd
from synfmc_amd import hip_ops as K
from tools.probe_gpu import bench
dev, dt = "cuda", torch.bfloat16
print("--- linear: ours vs F.linear (+bias)")
for M, Kd, N in [(81920, 320, 960), (81920, 320, 320), (81920, 320, 2560), (81920, 1280, 320), (20480, 640, 1920),
                 (20480, 640, 640), (20480, 640, 5120), (20480, 2560, 640), (5120, 1280, 3840), (5120, 1280, 1280),
                 (5120, 1280, 10240), (5120, 5120, 1280), (1280, 1280, 3840), (1280, 5120, 1280)]:
    x = torch.randn(M, Kd, device=dev, dtype=dt)
    w = torch.randn(N, Kd, device=dev, dtype=dt) * 0.02
    b = torch.zeros(N, device=dev, dtype=dt)
    r = torch.randn(M, N, device=dev, dtype=dt)
    t0 = bench(lambda: F.linear(x, w, b))
    t1 = bench(lambda: K.linear_bf16(x, w, b))
    t2 = bench(lambda: F.linear(x, w, b) + r)
    t3 = bench(lambda: K.linear_bf16(x, w, b, r))
    fl = 2.0 * M * Kd * N / 1e9
    print(f"M={M} K={Kd} N={N}: blaslt {t0:6.3f} ms {fl/t0:6.0f} TF | ours {t1:6.3f} ms {fl/t1:6.0f} TF || +res: blaslt+add {t2:6.3f} | ours fused {t3:6.3f}", flush=True)
print("--- GEGLU FF1: F.linear + geglu kernel vs fused")
from synfmc_amd.models.layers import interleave_geglu
for M, C in [(81920, 320), (20480, 640), (5120, 1280)]:
    x = torch.randn(M, C, device=dev, dtype=dt)
    w = torch.randn(8 * C, C, device=dev, dtype=dt) * 0.02
    b = torch.zeros(8 * C, device=dev, dtype=dt)
    wi, bi = interleave_geglu(w, b)
    t0 = bench(lambda: K.geglu(F.linear(x, w, b)))
    t1 = bench(lambda: K.linear_bf16(x, wi, bi, geglu=True))
    print(f"M={M} C={C}: blaslt+geglu {t0:6.3f} ms | fused {t1:6.3f} ms  ({2.0*M*C*8*C/1e9/t1:6.0f} TF)", flush=True)
print("--- conv3x3: ours vs MIOpen NHWC")
for n, ci, co, h, w_ in [(32, 320, 320, 40, 64), (32, 640, 640, 20, 32), (32, 1280, 1280, 10, 16), (32, 2560, 1280, 10, 16),
                         (32, 1920, 640, 20, 32), (32, 960, 320, 40, 64), (32, 320, 640, 20, 32), (32, 1280, 1280, 5, 8),
                         (32, 640, 320, 40, 64), (32, 1920, 1280, 10, 16), (32, 1280, 640, 20, 32)]:
    x = torch.randn(n, h, w_, ci, device=dev, dtype=dt)
    wt = (torch.randn(co, ci, 3, 3, device=dev, dtype=dt) * 0.02).contiguous(memory_format=torch.channels_last)
    b = torch.zeros(co, device=dev, dtype=dt)
    te = torch.randn(n, co, device=dev, dtype=dt)
    xn = x.permute(0, 3, 1, 2)
    t0 = bench(lambda: F.conv2d(xn, wt, b, 1, 1) + te[:, :, None, None])
    t1 = bench(lambda: K.conv3x3_bf16(x, wt, b, te))
    fl = 2.0 * n * h * w_ * ci * co * 9 / 1e9
    print(f"conv N={n} {ci}->{co} {h}x{w_}: miopen+temb {t0:6.3f} ms {fl/t0:6.0f} TF | ours fused {t1:6.3f} ms {fl/t1:6.0f} TF", flush=True)
