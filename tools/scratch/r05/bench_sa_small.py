"""Inner-level attention launches (d = 160): whole-K/V kernel vs the tiled one (FMC_SA_SMALL=0 in a second process)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../..")
from synfmc_amd import hip_ops as K

def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

H = 8
for (B, S, Skv, D) in [(32, 160, 160, 160), (32, 160, 77, 160), (32, 40, 40, 160), (32, 40, 77, 160), (32, 640, 640, 80)]:
    C = H * D
    g = torch.Generator(device="cuda").manual_seed(1)
    if S == Skv:
        qkv = torch.randn(B, S, 3 * C, device="cuda", generator=g).bfloat16()
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    else:
        q = torch.randn(B, S, C, device="cuda", generator=g).bfloat16()
        kv = torch.randn(2, Skv, 2 * C, device="cuda", generator=g).bfloat16()
        k, v = kv[..., :C], kv[..., C:]
    us = t(lambda: K.spatial_attention(q, k, v, H))
    gf = 4 * B * H * S * Skv * D / 1e9
    print(f"FMC_SA_SMALL={os.environ.get('FMC_SA_SMALL','1')} B={B} S={S} Skv={Skv} D={D}: {us:.1f} us  {gf/us*1e-3:.3f} PF/s")
