#!/usr/bin/env python
"""k-lockstep split (split_k = -3 / -(16 + S)) of the 8-phase kernel vs the other arms on the 10x16 / 5x8-level convolutions and the small-M
projections: correctness against arm 13 (plain grid) and graph-timed launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from synfmc_amd import hip_ops as K
torch.manual_seed(0)
dev = "cuda"


def conv_case(n, h, w, cin, cout, temb=False, res=False):
    x = torch.randn(n, h, w, cin, device=dev, dtype=torch.bfloat16)
    wt = (torch.randn(cout, cin, 3, 3, device=dev, dtype=torch.bfloat16) * 0.02).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device=dev, dtype=torch.bfloat16)
    te = torch.randn(n, cout, device=dev, dtype=torch.bfloat16) if temb else None
    r = torch.randn(n, h, w, cout, device=dev, dtype=torch.bfloat16) if res else None
    run = lambda tile, sk=1: K.conv3x3_bf16(x, wt, b, te, r, tile=tile, split_k=sk)
    return run, 2.0 * n * h * w * cout * 9 * cin


def lin_case(M, N, Kd, res=False):
    x = torch.randn(M, Kd, device=dev, dtype=torch.bfloat16)
    wt = torch.randn(N, Kd, device=dev, dtype=torch.bfloat16) * 0.02
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    r = torch.randn(M, N, device=dev, dtype=torch.bfloat16) if res else None
    run = lambda tile, sk=1: K.linear_bf16(x, wt, b, r, 1.0, tile=tile, split_k=sk)
    return run, 2.0 * M * N * Kd


cases = [("conv 32x10x16 1280->1280 temb", conv_case(32, 10, 16, 1280, 1280, temb=True)),
         ("conv 32x10x16 1280->1280 res", conv_case(32, 10, 16, 1280, 1280, res=True)),
         ("conv 32x10x16 2560->1280", conv_case(32, 10, 16, 2560, 1280)),
         ("conv 32x10x16 1920->1280", conv_case(32, 10, 16, 1920, 1280)),
         ("conv 32x10x16 640->1280", conv_case(32, 10, 16, 640, 1280)),
         ("conv 32x5x8 1280->1280 temb", conv_case(32, 5, 8, 1280, 1280, temb=True)),
         ("conv 32x5x8 2560->1280", conv_case(32, 5, 8, 2560, 1280)),
         ("conv 16x10x16 1280->1280", conv_case(16, 10, 16, 1280, 1280)),
         ("conv 16x20x32 640->640", conv_case(16, 20, 32, 640, 640)),
         ("conv 32x20x32 640->640", conv_case(32, 20, 32, 640, 640)),
         ("lin 5120x1280x5120 res", lin_case(5120, 1280, 5120, res=True)),
         ("lin 5120x1280x1280 res", lin_case(5120, 1280, 1280, res=True)),
         ("lin 1280x1280x5120 res", lin_case(1280, 1280, 5120, res=True)),
         ("lin 5120x3840x1280", lin_case(5120, 3840, 1280))]
for name, (run, flops) in cases:
    ref = run(13).float()
    line = [f"{name:34s}"]
    for label, tile, sk in (("13", 13, 1), ("141", 141, 1), ("269", 269, 1), ("lock", 397, 1), ("S2", 13, -18), ("S3", 13, -19), ("S4", 13, -20),
                            ("S5", 13, -21), ("S6", 13, -22), ("S8", 13, -24), ("S10", 13, -26)):
        try:
            out = run(tile, sk).float()
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            # twice: determinism of the split's summation order
            same = torch.equal(run(tile, sk), run(tile, sk))
            ms = K._time_ms(lambda: run(tile, sk))
            line.append(f"{label} {ms * 1e3:6.1f}us {flops / ms / 1e9:5.0f}TF e={err:.1e}{'' if same else ' NONDET'}")
        except Exception as e:
            line.append(f"{label} FAIL {str(e)[:40]}")
    print(" | ".join(line), flush=True)
