"""fmc_linear4_bf16 against the front-end's current choice on the inner levels' projection shapes."""
import sys, torch
sys.path.insert(0, ".")
from synfmc_amd import hip_ops as K
def t_ms(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
K.LINEAR4 = False
for M, N, Kd, res in [(5120, 1280, 1280, True), (5120, 1280, 1280, False), (5120, 3840, 1280, False), (5120, 1280, 5120, True), (5120, 2560, 768, False),
                      (1280, 1280, 1280, True), (1280, 3840, 1280, False), (1280, 1280, 5120, True), (20480, 640, 640, True), (20480, 1920, 640, False), (20480, 640, 2560, True)]:
    x = torch.randn(M, Kd, device="cuda").bfloat16(); w = (torch.randn(N, Kd, device="cuda") * Kd ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16(); r = torch.randn(M, N, device="cuda").bfloat16() if res else None
    a = t_ms(lambda: K.linear(x, w, b, r))
    arm = K._choice.get(("lin", M, N, Kd, True, int(res), 0))
    g = t_ms(lambda: K.linear4_bf16(x, w, b, r))
    v = t_ms(lambda: torch.nn.functional.linear(x, w, b))
    fl = 2.0 * M * N * Kd
    print(f"{M}x{N}x{Kd} res={int(res)}: front-end (arm {arm}) {a*1e3:6.1f} us | vendor GEMM alone {v*1e3:6.1f} | linear4 {g*1e3:6.1f} us = {fl/g/1e9:5.0f} TF/s ({fl/g/1e9/2500:.3f})", flush=True)
