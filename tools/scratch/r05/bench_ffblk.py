"""Second GEMM of the feed-forward: row-major intermediate through the tuned front-end vs the tile-major one (linear_from_blocked)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../..")
from synfmc_amd import hip_ops as K

def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

K.load_autotune_table()
for (M, N, Kd) in [(81920, 320, 1280), (20480, 640, 2560)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.randn(M, Kd, device="cuda", generator=g) * 0.5).bfloat16()
    w = (torch.randn(N, Kd, device="cuda", generator=g) * 0.03).bfloat16()
    b = torch.randn(N, device="cuda", generator=g).bfloat16()
    r = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    xb = x.view(M // 160, 160, Kd // 32, 32).permute(0, 2, 1, 3).contiguous().view(M, Kd)
    ref = K.linear(x, w, b, residual=r)
    got = K.linear_from_blocked(xb, w, b, r)
    err = (ref.float() - got.float()).abs().max().item() / ref.float().abs().max().item()
    us0 = t(lambda: K.linear(x, w, b, residual=r))
    us1 = t(lambda: K.linear_from_blocked(xb, w, b, r))
    print(f"M={M} N={N} K={Kd}: row-major {us0:.1f} us, tile-major {us1:.1f} us, rel diff {err:.2e}")
