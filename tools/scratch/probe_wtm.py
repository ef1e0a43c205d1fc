"""Tile 16 (row-major W) vs tile 18 (W pre-packed tile-major): projections, GEGLU, 3x3 convs of the three U-Net levels.  Bit-equality checked."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from synfmc_amd import hip_ops as K
from synfmc_amd.models.layers import interleave_geglu

torch.manual_seed(0)
bf = torch.bfloat16
for (M, N, Kd, res) in [(81920, 320, 320, True), (81920, 960, 320, False), (81920, 320, 1280, True), (20480, 640, 640, True), (20480, 1920, 640, False),
                        (20480, 640, 2560, True), (5120, 1280, 1280, True), (5120, 3840, 1280, False), (5120, 1280, 5120, True)]:
    x = torch.randn(M, Kd, device="cuda", dtype=bf)
    w = torch.randn(N, Kd, device="cuda", dtype=bf) * Kd ** -0.5
    b = torch.randn(N, device="cuda", dtype=bf)
    r = torch.randn(M, N, device="cuda", dtype=bf) if res else None
    same = torch.equal(K.linear_bf16(x, w, b, r, 1.0, tile=K.ARM_160), K.linear_bf16(x, w, b, r, 1.0, tile=K.ARM_160B))
    t16 = K._time_ms(lambda: K.linear_bf16(x, w, b, r, 1.0, tile=K.ARM_160), reps=10)
    t18 = K._time_ms(lambda: K.linear_bf16(x, w, b, r, 1.0, tile=K.ARM_160B), reps=10)
    print(f"lin {M}x{N}x{Kd} res={res}: row-major W {t16 * 1e3:7.1f} us   tile-major W {t18 * 1e3:7.1f} us  ({100 * (t18 / t16 - 1):+.1f} %)  equal={same}", flush=True)
for (M, N, Kd) in [(81920, 2560, 320), (20480, 5120, 640), (5120, 10240, 1280)]:
    x = torch.randn(M, Kd, device="cuda", dtype=bf)
    w = torch.randn(N, Kd, device="cuda", dtype=bf) * Kd ** -0.5
    b = torch.randn(N, device="cuda", dtype=bf)
    w8, b8 = interleave_geglu(w, b, 8)
    same = torch.equal(K.linear_bf16(x, w8, b8, geglu=True, tile=K.ARM_160), K.linear_bf16(x, w8, b8, geglu=True, tile=K.ARM_160B))
    t16 = K._time_ms(lambda: K.linear_bf16(x, w8, b8, geglu=True, tile=K.ARM_160), reps=10)
    t18 = K._time_ms(lambda: K.linear_bf16(x, w8, b8, geglu=True, tile=K.ARM_160B), reps=10)
    print(f"geglu {M}x{N}x{Kd}: row-major W {t16 * 1e3:7.1f} us   tile-major W {t18 * 1e3:7.1f} us  ({100 * (t18 / t16 - 1):+.1f} %)  equal={same}", flush=True)
for (n, h, w_, ci, co, kw) in [(32, 40, 64, 320, 320, {}), (32, 20, 32, 640, 640, {}), (32, 20, 32, 1280, 640, {}), (32, 10, 16, 1280, 1280, {}),
                               (32, 10, 16, 640, 640, {"upsample": True}), (32, 40, 64, 320, 320, {"stride2": True})]:
    hin, win = (h // 2, w_ // 2) if kw.get("upsample") else ((h * 2, w_ * 2) if kw.get("stride2") else (h, w_))
    x = torch.randn(n, hin, win, ci, device="cuda", dtype=bf)
    f = (torch.randn(co, ci, 3, 3, device="cuda", dtype=bf) * (9 * ci) ** -0.5).contiguous(memory_format=torch.channels_last)
    same = torch.equal(K.conv3x3_bf16(x, f, None, None, None, tile=K.ARM_160, **kw), K.conv3x3_bf16(x, f, None, None, None, tile=K.ARM_160B, **kw))
    t16 = K._time_ms(lambda: K.conv3x3_bf16(x, f, None, None, None, tile=K.ARM_160, **kw), reps=6)
    t18 = K._time_ms(lambda: K.conv3x3_bf16(x, f, None, None, None, tile=K.ARM_160B, **kw), reps=6)
    print(f"conv {n}x{h}x{w_} {ci}->{co} {kw}: row-major W {t16 * 1e3:7.1f} us   tile-major W {t18 * 1e3:7.1f} us  ({100 * (t18 / t16 - 1):+.1f} %)  equal={same}", flush=True)
