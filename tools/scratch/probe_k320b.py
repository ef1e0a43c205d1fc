#!/usr/bin/env python
"""The level-0 (K = 320 / 640) shapes of the step, us per call per arm (GPU box only). usage: probe_k320b.py [arms...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from synfmc_amd import hip_ops as K
from synfmc_amd.models.layers import interleave_geglu
dev, dt = "cuda", torch.bfloat16
arms = [int(a) for a in sys.argv[1:]] or [5, 11, 13, 15]
def t(fn): return K._time_ms(fn) * 1e3
for M, Kd, N, res in [(81920, 320, 320, True), (81920, 320, 320, False), (81920, 320, 960, False), (81920, 1280, 320, True), (20480, 640, 640, True),
                      (20480, 640, 1920, False), (20480, 2560, 640, True), (5120, 1280, 1280, True), (5120, 1280, 3840, False), (5120, 5120, 1280, True)]:
    a = torch.randn(M, Kd, device=dev, dtype=dt); w = torch.randn(N, Kd, device=dev, dtype=dt) * 0.02
    b = torch.zeros(N, device=dev, dtype=dt); r = torch.randn(M, N, device=dev, dtype=dt) if res else None
    ref = K.linear_bf16(a, w, b, r, tile=1)
    ok = all(torch.equal(K.linear_bf16(a, w, b, r, tile=x), ref) for x in arms)
    print(f"lin M={M} K={Kd} N={N} res={int(res)} equal={ok}: lib {t(lambda: (F.linear(a, w, b) + r) if res else F.linear(a, w, b)):6.1f} | " +
          " ".join(f"{x}:{t(lambda x=x: K.linear_bf16(a, w, b, r, tile=x)):6.1f}" for x in arms), flush=True)
for M, C in [(81920, 320), (20480, 640), (5120, 1280)]:
    a = torch.randn(M, C, device=dev, dtype=dt); w = torch.randn(8 * C, C, device=dev, dtype=dt) * 0.02
    wi, _ = interleave_geglu(w, None)
    print(f"geglu M={M} C={C}: lib {t(lambda: K.geglu(F.linear(a, w))):6.1f} | " + " ".join(f"{x}:{t(lambda x=x: K.linear_bf16(a, wi, None, geglu=True, tile=x)):6.1f}" for x in arms), flush=True)
for n, ci, co, h, w_ in [(32, 320, 320, 40, 64), (32, 640, 320, 40, 64), (32, 640, 640, 20, 32), (32, 1280, 1280, 10, 16)]:
    x = torch.randn(n, h, w_, ci, device=dev, dtype=dt)
    wt = (torch.randn(co, ci, 3, 3, device=dev, dtype=dt) * 0.02).contiguous(memory_format=torch.channels_last)
    r = torch.randn(n, h, w_, co, device=dev, dtype=dt)
    ref = K.conv3x3_bf16(x, wt, None, None, r, tile=1)
    ok = all(torch.equal(K.conv3x3_bf16(x, wt, None, None, r, tile=a_), ref) for a_ in arms)
    print(f"conv+res {n}x{h}x{w_} {ci}->{co} equal={ok}: " + " ".join(f"{a_}:{t(lambda a_=a_: K.conv3x3_bf16(x, wt, None, None, r, tile=a_)):6.1f}" for a_ in arms), flush=True)
