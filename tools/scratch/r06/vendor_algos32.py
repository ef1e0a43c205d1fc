"""hipBLASLt's heuristic list beyond its first 8 candidates (FMC_VENDOR_MAXALGOS=32): per main vendor shape of the step, every candidate's time, the best of the first 8 and the best overall."""
import os, torch
from synfmc_amd import hip_ops as K
from synfmc_amd import _lib
dev = torch.device("cuda:0"); dt = torch.bfloat16
lib = _lib.load()
K._vendor_init(0)
ws = K._vendor_workspace(dev)
shapes = [(5120, 1280, 1280, True, True), (5120, 1280, 1280, True, False), (5120, 3840, 1280, False, False), (5120, 1280, 5120, True, True), (5120, 1280, 2560, True, False),
          (1280, 1280, 1280, True, True), (1280, 1280, 5120, True, True), (20480, 640, 640, True, True), (20480, 1920, 640, False, False), (5120, 10240, 1280, True, False)]
tot8 = totall = 0.0
for (M, N, Kd, hb, hr) in shapes:
    x = torch.randn(M, Kd, device=dev, dtype=dt); w = torch.randn(N, Kd, device=dev, dtype=dt) * Kd ** -0.5
    b = torch.randn(N, device=dev, dtype=dt) if hb else None
    r = torch.randn(M, N, device=dev, dtype=dt) if hr else None
    out = torch.empty(M, N, device=dev, dtype=dt)
    n = lib.fmc_vendor_linear_candidates(M, N, Kd, Kd, N if hr else 0, N, int(hb), int(hr))
    times = []
    for a in range(n):
        call = lambda: lib.fmc_vendor_linear_bf16(x.data_ptr(), w.data_ptr(), K._p(b), K._p(r), out.data_ptr(), M, N, Kd, Kd, N if hr else 0, N, a, ws.data_ptr(), ws.numel(), K._stream())
        rc = call()
        if rc != 0:
            times.append(float("inf")); continue
        for _ in range(5): call()
        best = float("inf")
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): call()
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        times.append(best * 1e3)
    b8 = min(times[:8]); ball = min(times); ia = times.index(ball)
    tot8 += b8; totall += ball
    print(f"{M}x{N}x{Kd} bias={hb} res={hr}: {n} candidates, first {times[0]:.1f} us, best of 8 {b8:.1f} (#{times.index(b8)}), best of all {ball:.1f} (#{ia})")
print(f"sum best-of-8 {tot8:.1f} us, best-of-all {totall:.1f} us")
