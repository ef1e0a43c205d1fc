#!/usr/bin/env python
"""Knock-outs of gemm160_kernel's main loop (FMC_G160_DBG, one process per setting): conv 32x20x32 640->640 and linear 81920x320x1280."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from synfmc_amd import hip_ops as K
dev, bf = torch.device("cuda"), torch.bfloat16
x = torch.randn(32, 20, 32, 640, device=dev, dtype=bf)
f = (torch.randn(640, 640, 3, 3, device=dev, dtype=bf) * (9 * 640) ** -0.5).contiguous(memory_format=torch.channels_last)
t1 = K._time_ms(lambda: K.conv3x3_bf16(x, f, None, None, None, tile=512))
a = torch.randn(81920, 1280, device=dev, dtype=bf)
w = torch.randn(320, 1280, device=dev, dtype=bf) * 1280 ** -0.5
t2 = K._time_ms(lambda: K.linear_bf16(a, w, None, None, 1.0, tile=512))
a3 = torch.randn(20480, 640, device=dev, dtype=bf)
w3 = torch.randn(640, 640, device=dev, dtype=bf) * 640 ** -0.5
t3 = K._time_ms(lambda: K.linear_bf16(a3, w3, None, None, 1.0, tile=512))
print(f"dbg={os.environ.get('FMC_G160_DBG', '0'):>2s}  conv 640->640 @20x32: {t1 * 1e3:7.1f} us   lin 81920x320x1280: {t2 * 1e3:7.1f} us   lin 20480x640x640: {t3 * 1e3:6.1f} us", flush=True)
