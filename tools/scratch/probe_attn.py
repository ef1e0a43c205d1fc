"""Timing probe for the spatial attention forward kernel at the U-Net's shapes (bf16).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from synfmc_amd import hip_ops as K


def bench(fn, iters=20, warm=3):
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


def main():
    dev = "cuda"
    torch.manual_seed(0)
    print("FMC_SA_PREFETCH =", os.environ.get("FMC_SA_PREFETCH"))
    for (B, S, H, D) in [(32, 2560, 8, 40), (32, 640, 8, 80), (32, 160, 8, 160), (32, 40, 8, 160)]:
        C = H * D
        qkv = torch.randn(B, S, 3 * C, device=dev, dtype=torch.bfloat16)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        ms = bench(lambda: K._spatial_attention_raw(q, k, v, H))
        fl = 4.0 * B * H * S * S * D
        ref = torch.nn.functional.scaled_dot_product_attention(
            *(t.float().reshape(B, S, H, D).transpose(1, 2) for t in (q, k, v))).transpose(1, 2).reshape(B, S, C)
        out = K._spatial_attention_raw(q, k, v, H)
        out = out[0] if isinstance(out, tuple) else out
        err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
        print(f"self  B={B} S={S} d={D}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF/s  rel_inf={err:.2e}", flush=True)
        kt = torch.randn(2, 77, 2 * C, device=dev, dtype=torch.bfloat16)
        kk, vv = kt[..., :C], kt[..., C:]
        ms = bench(lambda: K._spatial_attention_raw(q, kk, vv, H))
        print(f"cross B={B} S={S} d={D}: {ms:7.3f} ms {4.0 * B * H * S * 77 * D / ms / 1e9:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
