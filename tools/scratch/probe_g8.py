#!/usr/bin/env python
"""8-phase GEMM arms on square shapes (GPU box only): plain grid vs stream-K, exact / inexact tile counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from synfmc_amd import hip_ops as K
dev, dt = "cuda", torch.bfloat16
arms = [int(a) for a in sys.argv[1:]] or [3, 13, 141]
for M, N, Kd in [(4096, 4096, 4096), (8192, 8192, 8192), (4096, 4096, 512), (5120, 4096, 4096), (20480, 2560, 2560), (20480, 1280, 11520)]:
    a = torch.randn(M, Kd, device=dev, dtype=dt)
    w = torch.randn(N, Kd, device=dev, dtype=dt) * 0.02
    fl = 2.0 * M * N * Kd
    print(f"M={M} N={N} K={Kd}: " + " ".join(f"{t}:{fl / K._time_ms(lambda t=t: K.linear_bf16(a, w, None, tile=t)) / 1e9:6.0f}" for t in arms), flush=True)
