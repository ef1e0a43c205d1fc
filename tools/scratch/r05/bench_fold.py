"""Camera-Adapter merge folded into the q | k | v projection (inner levels, un-fused temporal chain): two GEMMs vs one with a per-clip residual."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../..")
from synfmc_amd import hip_ops as K

def t(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

K.load_autotune_table()
C = 1280
with torch.no_grad():
    for M in (5120, 1280):
        g = torch.Generator(device="cuda").manual_seed(1)
        n_ = torch.randn(M, C, device="cuda", generator=g).bfloat16()
        pt = torch.randn(M, C, device="cuda", generator=g).bfloat16()
        wm = (torch.randn(C, C, device="cuda", generator=g) * 0.01).bfloat16()
        wq = (torch.randn(3 * C, C, device="cuda", generator=g) * 0.03).bfloat16()
        s = 1.0
        wf = (wq.float() @ (s * wm.float() + torch.eye(C, device="cuda"))).bfloat16()
        term = K.linear(pt, wq)
        def chain():
            m = K.linear(n_, wm, None, n_, s, residual2=pt)
            return K.linear(m, wq)
        def fold():
            return K.linear(n_, wf, None, term)
        a, b = chain(), fold()
        err = (a.float() - b.float()).abs().max().item() / a.float().abs().max().item()
        print(f"M={M}: merge + qkv {t(chain):.1f} us, folded {t(fold):.1f} us, rel diff {err:.2e}")
