#!/usr/bin/env python
"""Micro-probe of the building blocks at the 16x320x512 CFG-2 shapes (run on the GPU box):
MIOpen conv NHWC vs NCHW, hipBLASLt GEMMs, and the hand-written kernels.  Prints one line per case."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
import synfmc_amd
from synfmc_amd import hip_ops as K

dev, dt = "cuda", torch.bfloat16


def bench(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def conv_case(N, ci, co, h, w, k=3, stride=1):
    x = torch.randn(N, ci, h, w, device=dev, dtype=dt)
    wgt = torch.randn(co, ci, k, k, device=dev, dtype=dt) * 0.02
    b = torch.zeros(co, device=dev, dtype=dt)
    flops = 2.0 * N * (h // stride) * (w // stride) * ci * co * k * k
    res = {}
    for name, xx, ww in (("nchw", x, wgt), ("nhwc", x.contiguous(memory_format=torch.channels_last),
                                            wgt.contiguous(memory_format=torch.channels_last))):
        try:
            ms = bench(lambda: F.conv2d(xx, ww, b, stride, k // 2))
            res[name] = f"{ms:7.3f} ms {flops / ms / 1e9:7.1f} TF/s"
        except Exception as e:
            res[name] = f"ERR {e!r}"[:60]
    y = F.conv2d(x.contiguous(memory_format=torch.channels_last), wgt.contiguous(memory_format=torch.channels_last), b, stride, k // 2)
    print(f"conv{k}x{k} s{stride} N={N} {ci}->{co} {h}x{w}: nchw {res['nchw']} | nhwc {res['nhwc']} | out_cl={y.is_contiguous(memory_format=torch.channels_last)}", flush=True)


def gemm_case(M, Kd, N, bias=True):
    x = torch.randn(M, Kd, device=dev, dtype=dt)
    w = torch.randn(N, Kd, device=dev, dtype=dt) * 0.02
    b = torch.zeros(N, device=dev, dtype=dt) if bias else None
    ms = bench(lambda: F.linear(x, w, b))
    print(f"linear M={M} K={Kd} N={N}: {ms:7.3f} ms {2.0 * M * Kd * N / ms / 1e9:7.1f} TF/s", flush=True)


def main():
    print(torch.__version__, torch.cuda.get_device_name(0), "NHWC env", os.environ.get("PYTORCH_MIOPEN_SUGGEST_NHWC"))
    only_kernels = os.environ.get("PROBE", "") == "kernels"
    for bm in (() if only_kernels else (False, True)):
        torch.backends.cudnn.benchmark = bm
        print(f"--- cudnn.benchmark={bm}")
        for case in [(32, 320, 320, 40, 64), (32, 640, 640, 20, 32), (32, 1280, 1280, 10, 16), (32, 2560, 1280, 10, 16),
                     (32, 1920, 640, 20, 32), (32, 960, 320, 40, 64), (32, 320, 640, 20, 32), (32, 1280, 1280, 5, 8)]:
            conv_case(*case)
        conv_case(32, 320, 320, 40, 64, 3, 2)
    print("--- GEMM")
    for M, Kd, N in [] if only_kernels else [(81920, 320, 960), (81920, 320, 320), (81920, 320, 2560), (81920, 1280, 320), (20480, 640, 1920),
                     (20480, 640, 5120), (20480, 2560, 640), (5120, 1280, 3840), (5120, 1280, 10240), (5120, 5120, 1280),
                     (154, 768, 640)]:
        gemm_case(M, Kd, N)
    print("--- hand-written kernels (bf16)")
    for (B, S, H, D) in [(32, 2560, 8, 40), (32, 640, 8, 80), (32, 160, 8, 160), (32, 40, 8, 160)]:
        C = H * D
        qkv = torch.randn(B, S, 3 * C, device=dev, dtype=dt)
        ms = bench(lambda: K.spatial_attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], H))
        fl = 4.0 * B * H * S * S * D
        print(f"spatial_attn self B={B} S={S} d={D}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF/s", flush=True)
        q = torch.randn(B, S, C, device=dev, dtype=dt)
        kv = torch.randn(2, 77, 2 * C, device=dev, dtype=dt)
        ms = bench(lambda: K.spatial_attention(q, kv[..., :C], kv[..., C:], H))
        print(f"spatial_attn cross B={B} S={S} d={D}: {ms:7.3f} ms {4.0 * B * H * S * 77 * D / ms / 1e9:7.1f} TF/s", flush=True)
    for (B, P, H, D) in [(2, 2560, 8, 40), (2, 640, 8, 80), (2, 160, 8, 160), (2, 40, 8, 160)]:
        C = H * D
        qkv = torch.randn(B, 16, P, 3 * C, device=dev, dtype=dt)
        ms = bench(lambda: K.temporal_attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], H))
        by = 4.0 * B * 16 * P * C * 2
        print(f"temporal_attn B={B} P={P} d={D}: {ms:7.3f} ms {by / ms / 1e6:7.1f} GB/s", flush=True)
    for (N, HW, C) in [(32, 2560, 320), (32, 2560, 640), (32, 2560, 960), (32, 640, 640), (32, 640, 1920), (32, 160, 1280), (32, 160, 2560)]:
        x = torch.randn(N, HW, C, device=dev, dtype=dt)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        ms = bench(lambda: K.groupnorm_silu(x, g, b, 32, 1e-5, True))
        print(f"groupnorm_silu N={N} HW={HW} C={C}: {ms:7.3f} ms {2.0 * x.numel() * 2 / ms / 1e6:7.1f} GB/s", flush=True)
        ms = bench(lambda: F.silu(F.group_norm(x.permute(0, 2, 1), 32, g.to(dt), b.to(dt))))
        print(f"   torch group_norm+silu (NCL view): {ms:7.3f} ms", flush=True)
    x = torch.randn(81920, 320, device=dev, dtype=dt)
    g, b = torch.ones(320, device=dev), torch.zeros(320, device=dev)
    ms = bench(lambda: K.layernorm(x, g, b))
    print(f"layernorm M=81920 C=320: {ms:7.3f} ms {2.0 * x.numel() * 2 / ms / 1e6:7.1f} GB/s")
    x = torch.randn(81920, 2560, device=dev, dtype=dt)
    ms = bench(lambda: K.geglu(x))
    print(f"geglu M=81920 Cff=1280: {ms:7.3f} ms {1.5 * x.numel() * 2 / ms / 1e6:7.1f} GB/s")


if __name__ == "__main__":
    main()
