"""Small-M projections (the vendor arm's shapes): own arms incl. the 64 x 128 experiment (600 / 601) against F.linear (+ add)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import torch.nn.functional as F
from synfmc_amd import hip_ops as K

torch.manual_seed(0)
for (M, N, Kd, res) in [(1280, 1280, 1280, True), (1280, 3840, 1280, False), (1280, 1280, 5120, True), (5120, 1280, 1280, True), (5120, 3840, 1280, False),
                        (5120, 1280, 5120, True), (2560, 1280, 1280, True), (640, 1280, 1280, True)]:
    x = torch.randn(M, Kd, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, Kd, device="cuda", dtype=torch.bfloat16) * Kd ** -0.5
    b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    r = torch.randn(M, N, device="cuda", dtype=torch.bfloat16) if res else None
    ref = F.linear(x.float(), w.float(), b.float()) + (r.float() if res else 0)
    out = {}
    lib = (lambda: torch.add(r, F.linear(x, w, b))) if res else (lambda: F.linear(x, w, b))
    out["vendor"] = (K._time_ms(lib, reps=20), 0.0)
    for arm in (1, 2, 13, 600, 602, 603):
        try:
            y = K.linear_bf16(x, w, b, r, 1.0, tile=arm)
            e = ((y.float() - ref).abs().max() / ref.abs().max()).item()
            out[arm] = (K._time_ms(lambda: K.linear_bf16(x, w, b, r, 1.0, tile=arm), reps=20), e)
        except Exception as ex:
            out[arm] = (float("nan"), str(ex)[:40])
    print(f"{M}x{N}x{Kd} res={res}: " + " | ".join(f"{k} {v[0] * 1e3:6.1f}us e={v[1] if isinstance(v[1], str) else format(v[1], '.1e')}" for k, v in out.items()), flush=True)
