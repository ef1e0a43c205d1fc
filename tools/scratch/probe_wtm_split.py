"""Split-K arms of tile 16 on the 5x8 / 10x16 levels: row-major (513..516) vs tile-major weights (545..548)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from synfmc_amd import hip_ops as K

torch.manual_seed(0)
bf = torch.bfloat16
for (n, h, w_, ci, co) in [(32, 5, 8, 1280, 1280), (32, 5, 8, 2560, 1280), (32, 10, 16, 1280, 1280), (32, 10, 16, 640, 1280)]:
    x = torch.randn(n, h, w_, ci, device="cuda", dtype=bf)
    f = (torch.randn(co, ci, 3, 3, device="cuda", dtype=bf) * (9 * ci) ** -0.5).contiguous(memory_format=torch.channels_last)
    line = f"conv {n}x{h}x{w_} {ci}->{co}:"
    for si in (1, 2, 3):
        a, b = K.ARM_160 + si, K.ARM_160B + si
        same = torch.equal(K.conv3x3_bf16(x, f, None, None, None, tile=a), K.conv3x3_bf16(x, f, None, None, None, tile=b))
        ta = K._time_ms(lambda: K.conv3x3_bf16(x, f, None, None, None, tile=a), reps=8)
        tb = K._time_ms(lambda: K.conv3x3_bf16(x, f, None, None, None, tile=b), reps=8)
        line += f"  split {1 << si}: {ta * 1e3:6.1f} -> {tb * 1e3:6.1f} us ({100 * (tb / ta - 1):+.1f} %, eq={same})"
    print(line, flush=True)
for (M, N, Kd) in [(1280, 1280, 5120), (1280, 1280, 1280), (5120, 1280, 5120)]:
    x = torch.randn(M, Kd, device="cuda", dtype=bf)
    w = torch.randn(N, Kd, device="cuda", dtype=bf) * Kd ** -0.5
    line = f"lin {M}x{N}x{Kd}:"
    for si in (1, 2, 3):
        a, b = K.ARM_160 + si, K.ARM_160B + si
        ta = K._time_ms(lambda: K.linear_bf16(x, w, None, None, 1.0, tile=a), reps=8)
        tb = K._time_ms(lambda: K.linear_bf16(x, w, None, None, 1.0, tile=b), reps=8)
        line += f"  split {1 << si}: {ta * 1e3:6.1f} -> {tb * 1e3:6.1f} us ({100 * (tb / ta - 1):+.1f} %)"
    print(line, flush=True)
