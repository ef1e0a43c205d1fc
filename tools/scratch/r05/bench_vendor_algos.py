"""hipBLASLt heuristic candidates per shape of the step's vendor-arm projections: time of candidate 0 vs the best."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../..")
from synfmc_amd import hip_ops as K

def t(fn, n=60):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

lib = K._lib.load()
shapes = [(5120, 1280, 1280, 1, 1, 40), (5120, 3840, 1280, 0, 0, 10), (5120, 3840, 1280, 0, 1, 5), (5120, 1280, 5120, 1, 1, 10), (1280, 1280, 1280, 1, 1, 18), (5120, 1280, 1280, 1, 0, 10),
          (1280, 3840, 1280, 0, 0, 6), (1280, 3840, 1280, 0, 1, 5), (1280, 1280, 5120, 1, 1, 6), (5120, 1280, 1280, 0, 0, 5)]
tot0 = totb = 0.0
for (M, N, Kd, hb, hr, calls) in shapes:
    x = torch.randn(M, Kd, device="cuda").bfloat16(); w = (torch.randn(N, Kd, device="cuda") * 0.03).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16() if hb else None
    r = torch.randn(M, N, device="cuda").bfloat16() if hr else None
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    n = lib.fmc_vendor_linear_candidates(M, N, Kd, Kd, N if hr else 0, N, hb, hr)
    st = torch.cuda.current_stream().cuda_stream
    ts = []
    for a in range(n):
        ts.append(t(lambda: lib.fmc_vendor_linear_bf16(x.data_ptr(), w.data_ptr(), b.data_ptr() if hb else None, r.data_ptr() if hr else None, out.data_ptr(), M, N, Kd, Kd,
                                                        N if hr else 0, N, a, st)))
    best = min(range(n), key=lambda i: ts[i])
    tot0 += calls * ts[0]; totb += calls * ts[best]
    print(f"{(M, N, Kd)} bias={hb} res={hr} x{calls}: " + " ".join(f"{v:.1f}" for v in ts) + f"  -> best {best}")
print(f"per step: candidate 0 {tot0 / 1e3:.3f} ms, best {totb / 1e3:.3f} ms")
