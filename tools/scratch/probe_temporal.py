#!/usr/bin/env python
"""The level-0 temporal attention launch of the benchmarked step (bf16 and fp8 forms) and the roofline conv on the 8-phase arm, a few
times each, for rocprofv3 PMC passes (tools/pmc_temporal.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from synfmc_amd import hip_ops as K
B, Fr, P, H, D = 2, 16, 2560, 8, 40
C = H * D
qkv = torch.randn(B, Fr, P, 3 * C, device="cuda", dtype=torch.bfloat16)
q8 = torch.randn(B, Fr, P, 3 * C, device="cuda").clamp(-3, 3).to(torch.float8_e4m3fn)
sc = torch.ones(3, device="cuda")
x = torch.randn(32, 20, 32, 640, device="cuda", dtype=torch.bfloat16)
wt = (torch.randn(640, 640, 3, 3, device="cuda", dtype=torch.bfloat16) * 0.02).contiguous(memory_format=torch.channels_last)
for _ in range(6):
    K.self_attention_qkv(qkv, H, D ** -0.5, True)
    K._temporal_fp8_raw(q8, sc, H, D ** -0.5)
    K.conv3x3_bf16(x, wt, None, None, None, tile=13)
torch.cuda.synchronize()
