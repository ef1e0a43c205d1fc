"""LayerNorm / GroupNorm launches of the inner levels (latency-bound sizes): kernel time under rocprofv3."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../..")
from synfmc_amd import hip_ops as K
import torch.nn.functional as F
torch.manual_seed(0)
with torch.no_grad():
    for (M, C) in [(5120, 1280), (1280, 1280), (20480, 640), (81920, 320)]:
        x = torch.randn(M, C, device="cuda").bfloat16(); g = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
        for _ in range(50):
            y = K.layernorm(x, g, b, 1e-5)
        ref = F.layer_norm(x.float(), (C,), g, b, 1e-5)
        print("ln", M, C, float((y.float() - ref).abs().max()))
    for (N, HW, C) in [(32, 160, 1280), (32, 40, 1280), (32, 640, 640), (32, 160, 2560)]:
        x = torch.randn(N, C, HW // 8, 8, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        w = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
        for _ in range(50):
            y = K.groupnorm_silu(x, w, b, 32, 1e-5, True)
        print("gn", N, HW, C)
