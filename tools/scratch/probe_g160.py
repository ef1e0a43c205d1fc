#!/usr/bin/env python
"""Arm 16 (160 x 320 tiles, 8-phase schedule) against the other plain arms on the U-Net's level-0 / level-1 shapes: graph-timed launches
(hip_ops._time_ms), TFLOP/s, random data."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from synfmc_amd import hip_ops as K
from synfmc_amd.models.layers import interleave_geglu

dev = torch.device("cuda")
ARMS = [3, 5, 11, 13, 15, 512]
bf = torch.bfloat16


def lin(M, N, Kd, bias=True, res=False, geglu=False):
    x = torch.randn(M, Kd, device=dev, dtype=bf)
    w = torch.randn(N, Kd, device=dev, dtype=bf) * Kd ** -0.5
    b = torch.randn(N, device=dev, dtype=bf) if bias else None
    r = torch.randn(M, N, device=dev, dtype=bf) if res else None
    out = {}
    if geglu:
        w32, b32 = interleave_geglu(w, b)
        w160, b160 = interleave_geglu(w, b, 8)
    for t in ARMS:
        if t == 15 and (geglu or Kd != 320 or N % 320):
            continue
        if geglu:
            fn = (lambda: K.linear_bf16(x, w160, b160, geglu=True, tile=512)) if t == 512 else (lambda t=t: K.linear_bf16(x, w32, b32, geglu=True, tile=t))
        else:
            fn = lambda t=t: K.linear_bf16(x, w, b, r, 1.0, tile=t)
        out[t] = K._time_ms(fn)
    fl = 2.0 * M * N * Kd
    best = min(out, key=out.get)
    print(f"{'geglu' if geglu else 'lin':5s} M={M:6d} N={N:5d} K={Kd:5d} res={int(res)}  " + "  ".join(f"{t}:{v * 1e3:7.1f}us" for t, v in out.items())
          + f"   arm16 {fl / out[512] / 1e9:6.0f} TF/s, best other {fl / min(v for t, v in out.items() if t != 512) / 1e9:6.0f} (arm {best})", flush=True)


def conv(n, h, w_, ci, co, temb=False, res=False, ups=False):
    x = torch.randn(n, h // (2 if ups else 1), w_ // (2 if ups else 1), ci, device=dev, dtype=bf)
    f = (torch.randn(co, ci, 3, 3, device=dev, dtype=bf) * (9 * ci) ** -0.5).contiguous(memory_format=torch.channels_last)
    t_ = torch.randn(n, co, device=dev, dtype=bf) if temb else None
    r = torch.randn(n, h, w_, co, device=dev, dtype=bf) if res else None
    out = {}
    for t in ARMS:
        if t == 15:
            continue
        out[t] = K._time_ms(lambda t=t: K.conv3x3_bf16(x, f, None, t_, r, tile=t, upsample=ups))
    fl = 2.0 * n * h * w_ * 9 * ci * co
    best = min(out, key=out.get)
    print(f"conv  {n}x{h}x{w_} {ci:4d}->{co:4d} temb={int(temb)} res={int(res)} ups={int(ups)}  " + "  ".join(f"{t}:{v * 1e3:7.1f}us" for t, v in out.items())
          + f"   arm16 {fl / out[512] / 1e9:6.0f} TF/s, best other {fl / min(v for t, v in out.items() if t != 512) / 1e9:6.0f} (arm {best})", flush=True)


def small():
    global ARMS
    ARMS = [1 + 32, 2 + 32, 128 + 13, 256 + 13, 513, 514, 515]
    x = torch.randn(32, 5, 8, 1280, device=dev, dtype=bf)
    f = (torch.randn(1280, 1280, 3, 3, device=dev, dtype=bf) * (9 * 1280) ** -0.5).contiguous(memory_format=torch.channels_last)
    x2 = torch.randn(32, 5, 8, 2560, device=dev, dtype=bf)
    f2 = (torch.randn(1280, 2560, 3, 3, device=dev, dtype=bf) * (9 * 2560) ** -0.5).contiguous(memory_format=torch.channels_last)
    x3 = torch.randn(32, 10, 16, 1280, device=dev, dtype=bf)
    for name, xx, ff in (("conv 5x8 1280->1280", x, f), ("conv 5x8 2560->1280", x2, f2), ("conv 10x16 1280->1280", x3, f)):
        print(name, "  ".join(f"{t}:{K._time_ms(lambda t=t: K.conv3x3_bf16(xx, ff, None, None, None, tile=t)) * 1e3:7.1f}us" for t in ARMS), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "small":
        small()
        sys.exit(0)
    conv(32, 40, 64, 320, 320, temb=True)
    conv(32, 40, 64, 320, 320, res=True)
    conv(32, 40, 64, 640, 320, temb=True)
    conv(32, 40, 64, 960, 320, temb=True)
    conv(32, 40, 64, 640, 640, ups=True)
    conv(32, 20, 32, 640, 640, temb=True)
    conv(32, 20, 32, 640, 640, res=True)
    conv(32, 20, 32, 1280, 640, temb=True)
    conv(32, 20, 32, 1920, 640, temb=True)
    conv(32, 20, 32, 1280, 1280, ups=True)
    conv(32, 10, 16, 1280, 1280, temb=True)
    lin(81920, 320, 320, res=True)
    lin(81920, 320, 320)
    lin(81920, 960, 320, bias=False)
    lin(81920, 320, 1280, res=True)
    lin(81920, 320, 960)
    lin(81920, 2560, 320, geglu=True)
    lin(20480, 640, 640, res=True)
    lin(20480, 640, 640)
    lin(20480, 1920, 640, bias=False)
    lin(20480, 640, 2560, res=True)
    lin(20480, 5120, 640, geglu=True)
    lin(5120, 1280, 1280, res=True)
    lin(5120, 3840, 1280, bias=False)
    lin(5120, 1280, 5120, res=True)
    lin(5120, 10240, 1280, geglu=True)
