#!/usr/bin/env python
"""Streaming-bandwidth reference points (graph-timed): a plain copy vs the HBM-bound fmc kernels at the U-Net's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from synfmc_amd import hip_ops as K
dev, dt = "cuda", torch.bfloat16
for M, C in [(81920, 320), (20480, 640), (5120, 1280)]:
    x = torch.randn(M, C, device=dev, dtype=dt)
    y = torch.empty_like(x)
    r = torch.randn(M, C, device=dev, dtype=dt)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    nb = 2 * x.numel() * 2
    # rotate over several buffers so the reads really come from HBM, not the 256 MiB Infinity Cache
    xs = [torch.randn(M, C, device=dev, dtype=dt) for _ in range(8)]
    it = [0]
    def rot(fn):
        def f():
            it[0] = (it[0] + 1) % 8
            return fn(xs[it[0]])
        return f
    t_copy = K._time_ms(rot(lambda a: y.copy_(a)))
    t_add = K._time_ms(rot(lambda a: torch.add(a, r, out=y)))
    t_ln = K._time_ms(rot(lambda a: K._layernorm_raw(a.view(32, -1, C), g, b, 1e-5, None, 1, 1)))
    x4 = [a.view(32, -1, C) for a in xs]
    t_gn = K._time_ms(rot(lambda a: K.groupnorm_silu_raw(a.view(32, -1, C), g, b, 32, 1e-5, True)))
    print(f"M={M} C={C}: copy {nb/t_copy/1e6:6.0f} GB/s ({t_copy*1e3:5.1f} us) | add(3 streams) {1.5*nb/t_add/1e6:6.0f} GB/s ({t_add*1e3:5.1f} us)"
          f" | layernorm {nb/t_ln/1e6:6.0f} GB/s ({t_ln*1e3:5.1f} us) | groupnorm+silu {nb/t_gn/1e6:6.0f} GB/s alg ({t_gn*1e3:5.1f} us)", flush=True)
