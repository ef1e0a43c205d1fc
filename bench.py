#!/usr/bin/env python
"""bench.py -- denoising steps/sec of the FMC hot path on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one denoising step of `CameraObjCtrlPipeline`: the full-width 3-D U-Net (SD-1.5 layout + AnimateDiff
motion modules, 1.39 B params) forward at classifier-free-guidance batch 2 on a 16x320x512 clip with Camera
Adapter features (CMC) and Object-Motion-Control features (OMC) injected, plus the fused CFG + DDIM update
(reference: fmc/pipelines/pipeline_animation_cm_om.py:679-720).  Weights are random-init of that architecture,
inputs are the seeded synthetic clip of SURVEY.md section 8d ("data": "synthetic"); inputs are resident in HBM when
the timed region starts.  Camera encoder / OMC adapter run once per clip, outside the loop, exactly as in the
reference pipeline (their time is reported separately in the JSON).

Multi-GPU: clips are independent, so rank r denoises its own clip with no data-path collective ("scaling": "weak");
the timed region is bracketed by barrier + synchronize and the MAX over ranks is used.  `python bench.py --gpus N` with
no WORLD_SIZE in the environment starts the N ranks itself (re-exec under `torch.distributed.run`, 127.0.0.1 rendezvous);
under an external launcher WORLD_SIZE must equal --gpus.  `n_gpus` in the JSON is the world size an RCCL all-reduce of
ones returned, not an argument echo.

Before anything is timed, ONE step of the benchmarked model is compared with the CPU oracle (same bf16-rounded weights,
same noise / timestep / text / camera poses / object masks): `parity_rel_inf` in the JSON line.  The same oracle forward
-- a CFG-batch-2 16x320x512 U-Net + CMC + OMC step on the host cores -- is the `cpu_baseline` sample.

The JSON line also carries
  roofline     -- the dominant hand-written kernel (level-0 spatial self-attention, S=2560, d=40, 256 (batch,head)
                  pairs): algorithmic flops 4*B*H*S^2*d divided by its average launch duration measured here with
                  events on the launch stream; peak = 2.5 PFLOP/s dense bf16 MFMA (MI355X_MICROARCH.md);
  roofline_temporal -- the level-0 temporal attention kernel (HBM-bound): algorithmic bytes 4*N*F*H*D*2 / launch time
                  against 8 TB/s;
  roofline_temporal_block -- the FUSED temporal attention block of a 40x64-level motion module (LayerNorm + Camera-Adapter merge +
                  q | k | v + attention over the frames + out-projection in one launch): (merge + QKV + core + out) flops / launch time
                  against the 2.5 PFLOP/s dense bf16 MFMA peak -- the north-star's "MFMA utilisation on temporal attention";
  roofline_temporal_block_l1 -- the same block at the 20x32 level (C = 640, 8 heads x 80: its own kernel, temporal_block640.hip);
  cpu_baseline -- the oracle (fp32 PyTorch restatement; the reference itself needs diffusers, not installable)
                  timed on this node's host cores on ONE real step of the metric's configuration (no extrapolation).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

WIDTHS = (320, 640, 1280, 1280)
FRAMES, HEIGHT, WIDTH = 16, 320, 512
CROSS_DIM = 768
PEAK_BF16_TFLOPS = 2500.0          # dense MFMA bf16, MI355X_MICROARCH.md "Chip-level parameters"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def fast_init_(module: torch.nn.Module, seed: int, std: float = 0.02):
    """Seeded random init on the module's device, Kaiming-like as SURVEY.md section 8d prescribes: matrices / filters
    N(0, 1/fan_in) (variance preserving, so the parity figure measures arithmetic and not a chaotic net), vectors N(0, std),
    norm gains around 1.  Zero-initialised layers of the reference (qkv_merge, zero convs, LoRA up, proj_out) get values too
    so the conditioning paths do real work."""
    g = torch.Generator(device=next(module.parameters()).device).manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            p.normal_(0.0, std if p.ndim < 2 else p[0].numel() ** -0.5, generator=g)
            if p.ndim == 1 and "norm" in name and name.endswith("weight"):
                p.add_(1.0)


CONFIGS = {
    # BASELINE.json configs[3] -- the metric's workload (configs/obj.yaml)
    "obj": {"metric": "denoising steps/sec, 16x320x512 bf16 U-Net+CMC+OMC",
            "workload": "16x320x512 clip, CFG batch 2, full-width 3D U-Net (1.39B params, random init) + Camera Adapter (CMC) + "
                        "Object Motion Control (OMC) features, DDIM step; configs/obj.yaml shapes; 1 clip per GPU"},
    # configs[2] (configs/cam.yaml): UNet3DConditionModelPoseCond + CameraPoseEncoder, no OMC
    "cam": {"metric": "denoising steps/sec, 16x320x512 bf16 U-Net+CMC (configs/cam.yaml)",
            "workload": "16x320x512 clip, CFG batch 2, full-width 3D U-Net + Camera Encoder / Adapter with Pluecker rays (CMC only, "
                        "no object conditioning), DDIM step; configs/cam.yaml shapes; 1 clip per GPU"},
    # configs[1] (configs/lora.yaml): base U-Net + Domain LoRA, plain AnimationPipeline loop
    "lora": {"metric": "denoising steps/sec, 16x320x512 bf16 U-Net+Domain-LoRA (configs/lora.yaml, 50-step DDIM loop)",
             "workload": "16x320x512 clip, CFG batch 2, full-width 3D U-Net with the Domain LoRA on every spatial attention "
                         "(merged at load), no camera / object conditioning; steps of the 50-step DDIM loop of AnimationPipeline; "
                         "configs/lora.yaml shapes; 1 clip per GPU"},
}


def build_models(device, dtype, config="obj"):
    """(unet, camera encoder | None, OMC adapter | None) of one BASELINE configuration, seeded random init at full width."""
    from synfmc_amd.adapter import Adapter
    from synfmc_amd.models.pose_adaptor import CameraPoseEncoder
    from synfmc_amd.models.unet import UNet3DConditionModel, UNet3DConditionModelCamObjCond, UNet3DConditionModelPoseCond
    from synfmc_amd.modified_modules import patch_unet_for_omc
    from synfmc_amd import configs as CM
    enc = ada = None
    with torch.device(device):
        if config == "lora":
            unet = UNet3DConditionModel(**CM.unet_kwargs(WIDTHS, CROSS_DIM))
            unet.set_image_layer_lora(2)                                  # rank C / 2, fmc/models/unet.py:407-421, configs/lora.yaml
        else:
            cls = UNet3DConditionModelCamObjCond if config == "obj" else UNet3DConditionModelPoseCond
            unet = cls(**CM.unet_kwargs(WIDTHS, CROSS_DIM))
            unet.set_all_attn_processor(**CM.processor_kwargs(WIDTHS))
            enc = CameraPoseEncoder(**CM.encoder_kwargs(WIDTHS, max(16, FRAMES)))
            if config == "obj":
                ada = Adapter(**CM.adapter_kwargs(WIDTHS))
    if config == "obj":
        patch_unet_for_omc(unet)
    for i, m in enumerate((unet, enc, ada)):
        if m is not None:
            fast_init_(m, 1234 + i)
            m.to(dtype=dtype).eval().requires_grad_(False)
    return unet, enc, ada


def synthetic_inputs(rank, device):
    from synfmc_amd import configs as CM
    clip = CM.synthetic_clip(B=1, Fr=FRAMES, H=HEIGHT, W=WIDTH, n_obj=3, cross_dim=CROSS_DIM, seed=1234 + rank)
    g = torch.Generator().manual_seed(99 + rank)
    uncond = torch.randn(1, 77, CROSS_DIM, generator=g)
    return clip, torch.cat([uncond, clip["text"]]).to(device)


# Kernel sources whose hash keys the recorded hardware counters (profiles/roofline_counters.json, written by
# tools/collect_roofline_counters.py on a GPU box): a counter is printed only while the kernel it was collected on is the kernel
# in this tree -- it cannot silently go stale with the next kernel change.
_KERNEL_SOURCES = {"conv_halo_l0": ("conv_halo.hip", "common.h"), "conv_halo_l1": ("conv_halo.hip", "common.h"),
                   "sa40d": ("spatial_attn.hip", "attn_common.h", "common.h"),
                   "temporal": ("temporal_attn.hip", "attn_common.h", "common.h"),
                   "conv": ("gemm_conv.hip", "common.h"),
                   "proj": ("temporal_block640.hip", "gemm_conv.hip", "common.h"),
                   "tblock": ("temporal_block.hip", "common.h"),
                   "tblock640": ("temporal_block640.hip", "common.h"),
                   "proj_l0": ("geglu_pipe.hip", "common.h"), "ff2": ("gemm_conv.hip", "common.h"), "conv_halo4": ("conv_halo4.hip", "common.h")}


def kernel_source_sha(kernel: str) -> str:
    import hashlib
    h = hashlib.sha256()
    for f in _KERNEL_SOURCES[kernel]:
        with open(os.path.join(ROOT, "synfmc_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def recorded_counters(kernel: str) -> dict:
    """{"traffic": bytes | None, ...} for one roofline kernel from profiles/roofline_counters.json; None + a reason when the file is
    missing or was collected on other kernel sources."""
    path = os.path.join(ROOT, "profiles", "roofline_counters.json")
    try:
        with open(path) as f:
            rec = json.load(f)["kernels"][kernel]
    except (OSError, KeyError, ValueError):
        return {"traffic": None, "traffic_note": "no recorded counters (tools/collect_roofline_counters.py)"}
    if rec.get("source_sha16") != kernel_source_sha(kernel):
        return {"traffic": None, "traffic_note": f"recorded counters are for kernel sources {rec.get('source_sha16')}, this tree is "
                                                 f"{kernel_source_sha(kernel)}: re-run tools/collect_roofline_counters.py"}
    out = {"traffic": rec["traffic_bytes"], "traffic_source": "profiles/roofline_counters.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, "
                                                              "2 x FETCH + WRITE, collected on these kernel sources)"}
    for k in ("matrix_pipe_busy", "shader_clock_ghz_under_counters"):
        if k in rec:
            out[k] = rec[k]
    return out


def measure_attention_roofline(device, dtype, iters=20, clips=2):
    """Level-0 spatial self-attention exactly as the U-Net launches it: q/k/v are slices of one fused [32, 2560, 960]
    projection (CFG batch 2 x 16 frames, 8 heads x 40; `clips` = 1 for a training step)."""
    from synfmc_amd import hip_ops as K
    B, S, H, D = clips * FRAMES, (HEIGHT // 8) * (WIDTH // 8), 8, WIDTHS[0] // 8
    C = H * D
    qkv = torch.randn(B, S, 3 * C, device=device, dtype=dtype)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    for _ in range(3):
        K.spatial_attention(q, k, v, H)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        K.spatial_attention(q, k, v, H)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 4.0 * B * H * S * S * D
    achieved = flops / (ms * 1e-3) / 1e12
    out = {"bound": "mfma", "kernel": f"sa40d_kernel (software-pipelined spatial self-attention, bf16, d=40) [B*H={B * H},S={S}]", "achieved": round(achieved, 2),
           "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
           "avg_launch_ms": round(ms, 4), "flops_per_launch": flops, "traffic_algorithmic": 4.0 * B * S * H * D * 2}
    # HBM-side bytes per launch of this exact shape, matrix-pipe duty and shader clock: hardware counters, recorded by
    # tools/collect_roofline_counters.py (separate rocprofv3 --pmc passes) and only quoted for the kernel sources they were taken on
    if (B, S) == (32, 2560):                          # (the counters were collected on the metric's launch shape)
        out.update(recorded_counters("sa40d"))
    out["_match"] = ("name", "sa40d_kernel", "maxgrid")
    return out


def _time_launch(run, iters=20, warm=3):
    for _ in range(warm):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def measure_attention_level_roofline(device, dtype, level, iters=50):
    """Spatial self-attention of the 20x32 (`level` 1: 8 heads x 80, S = 640, `sa_big80_kernel`) or 10x16 level (2: 8 heads x 160, S = 160, `sa_small160_kernel`)
    as the U-Net launches it: q | k | v slices of one fused projection, CFG batch 2 x 16 frames.  MFMA roofline like level 0; at level 2 a launch is 4.2 GF on
    52 MB behind a launch + round-trip floor, so `frac` there measures latency, not the matrix pipe."""
    from synfmc_amd import hip_ops as K
    B, S, H, D = 2 * FRAMES, (HEIGHT // (8 << level)) * (WIDTH // (8 << level)), 8, WIDTHS[level] // 8
    C = H * D
    qkv = torch.randn(B, S, 3 * C, device=device, dtype=dtype)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    ms = _time_launch(lambda: K.spatial_attention(q, k, v, H), iters=iters)
    flops = 4.0 * B * H * S * S * D
    achieved = flops / (ms * 1e-3) / 1e12
    name = "sa_big80_kernel" if level == 1 else "sa_small160_kernel<5>"
    what = ("320-key tiles resident in LDS, 10 waves = 320 query rows" if level == 1 else "all keys resident in LDS, exact softmax, one round trip")
    out = {"bound": "mfma", "kernel": f"{name} ({what}; spatial self-attention, bf16, d={D}) [B*H={B * H},S={S}]", "achieved": round(achieved, 2),
           "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "avg_launch_ms": round(ms, 4), "flops_per_launch": flops,
           "traffic_algorithmic": 4.0 * B * S * H * D * 2, "traffic": None}
    out["_match"] = ("name", name)
    return out


def measure_attention_bwd_roofline(device, dtype, iters=10):
    """Backward of the level-0 spatial self-attention of a training step (one clip: 16 frames x 8 heads, S = h w, d = 40) exactly as autograd
    issues it: `fmc_spatial_attn_bwd` = rowdot + dQ pass + dK / dV pass (flash-style recompute from the forward's log-sum-exp).  Algorithmic flops
    2.5 x the forward's 4 B H S^2 D (recompute S, dP, dQ, dK, dV: five S x S x D products against the forward's two)."""
    from synfmc_amd import hip_ops as K
    B, S, H, D = FRAMES, (HEIGHT // 8) * (WIDTH // 8), 8, WIDTHS[0] // 8
    C = H * D
    qkv = torch.randn(B, S, 3 * C, device=device, dtype=dtype)
    q, k, v = (qkv[..., i * C:(i + 1) * C].detach().requires_grad_(True) for i in range(3))
    with torch.enable_grad():
        o = K.spatial_attention(q, k, v, H)
    do = torch.randn_like(o)

    def run():
        torch.autograd.grad(o, (q, k, v), do, retain_graph=True)
    ms = _time_launch(run, iters)
    flops = 2.5 * 4.0 * B * H * S * S * D
    achieved = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": f"fmc_spatial_attn_bwd (rowdot_kernel + attn_dq_kernel + attn_dkdv_kernel, bf16, d={D}) [B*H={B * H},S={S}]",
            "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
            "avg_launch_ms": round(ms, 4), "flops_per_launch": flops, "traffic": None,
            "note": "three launches per call, isolated loop (no in-step trace in train mode)"}


def measure_conv_roofline(device, dtype, level=1, iters=20):
    """The 3x3 convolutions are the largest block of the step: one of their launches exactly as the U-Net issues it -- `level` 1: ResNet conv of the
    20x32 level (CFG batch 2 x 16 frames, 640 -> 640), `level` 0: of the 40x64 level (320 -> 320) -- through the same front-end
    (`hip_ops.conv3x3`), i.e. on the kernel the dispatch picks for it (reported).  Algorithmic flops 2 M Cout 9 Cin; algorithmic bytes x + w + out."""
    from synfmc_amd import hip_ops as K
    d = 16 if level == 1 else 8
    n, h, w, ci, co = 2 * FRAMES, HEIGHT // d, WIDTH // d, WIDTHS[level], WIDTHS[level]
    x = torch.randn(n, h, w, ci, device=device, dtype=dtype).permute(0, 3, 1, 2)        # logical NCHW over channels-last storage
    wt = (torch.randn(co, ci, 3, 3, device=device, dtype=dtype) * 0.02).contiguous(memory_format=torch.channels_last)
    halo0 = K.conv_halo_calls["conv"]
    ms = _time_launch(lambda: K.conv3x3(x, wt, None), iters)
    halo = K.conv_halo_calls["conv"] > halo0
    arm = "halo" if halo else K._choice.get(("conv", n, h, w, ci, co, False, False, False, False))
    flops = 2.0 * n * h * w * 9 * ci * co
    achieved = flops / (ms * 1e-3) / 1e12
    names = {"halo": "conv_halo_kernel<plain> (input halo resident in LDS, W-only LDS-DMA stream, 10x32 pixel x 160 channel tiles)",
             0: "vendor library", 3: "gemm_kernel<conv3x3,256x256,16 waves>", 13: "gemm8_kernel<conv3x3,256x256,8-phase>",
             141: "gemm8_kernel<conv3x3,8-phase,stream-K>"}
    out = {"bound": "mfma", "kernel": f"{names.get(arm, f'fmc_conv3x3_bf16 arm {arm}')} [{n}x{h}x{w}, {ci}->{co}]",
           "autotuned_arm": arm, "achieved": round(achieved, 2),
           "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
           "avg_launch_ms": round(ms, 4), "flops_per_launch": flops,
           "traffic_algorithmic": 2.0 * (n * h * w * (ci + co) + 9 * ci * co),
           "_match": ("conv_halo", (n, h, w, ci, co, False)) if halo else None}
    out.update(recorded_counters("conv_halo_l%d" % level if halo else "conv"))
    return out


def measure_groupnorm_roofline(device, dtype, iters=30):
    """GroupNorm(32) + SiLU of the 40x64 level as the step runs it behind a halo convolution: the statistics arrive from the producing conv's
    epilogue, the launch is the apply pass alone (`fmc_groupnorm_apply_fwd`: read x, write silu(norm(x))).  HBM bound: 2 N HW C e bytes."""
    from synfmc_amd import hip_ops as K
    n, hw, C = 2 * FRAMES, (HEIGHT // 8) * (WIDTH // 8), WIDTHS[0]
    x = torch.randn(n, hw, C, device=device, dtype=dtype)
    g, b = torch.randn(C, device=device) * 0.2 + 1, torch.randn(C, device=device) * 0.1
    part = K.groupnorm_partials(x, 32)
    ms = _time_launch(lambda: K.groupnorm_apply(x, g, b, 32, 1e-5, True, part), iters)
    nbytes = 2.0 * n * hw * C * 2
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": f"gn_apply_fwd_kernel<bf16> (GroupNorm + SiLU apply pass, statistics from the producer's epilogue) [{n}x{hw}x{C}]",
            "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4), "avg_launch_ms": round(ms, 5),
            "bytes_per_launch": nbytes, "traffic": None, "_match": ("name", "gn_apply_fwd_kernel", "grid", 12960 if (hw, C) == (2560, 320) else -1)}


def measure_proj_roofline(device, dtype, iters=20):
    """One launch of the feed-forward input projection of the 20x32 level as the U-Net issues it since round 4: LayerNorm + GEGLU projection
    `[2F * 640 tokens, 640] x [5120, 640]^T` gated to 2560 columns with the A operand resident (`fmc_geglu640_ln_bf16`); where that kernel does not apply
    (other widths) the 160x320 GEGLU kernel through the autotuned front-end.  Timed like the other roofline launches."""
    from synfmc_amd import hip_ops as K
    from synfmc_amd.models.layers import interleave_geglu
    M, Kd, N = 2 * FRAMES * (HEIGHT // 16) * (WIDTH // 16), WIDTHS[1], 8 * WIDTHS[1]
    x = torch.randn(M, Kd, device=device, dtype=dtype)
    w = torch.randn(N, Kd, device=device, dtype=dtype) * Kd ** -0.5
    b = torch.randn(N, device=device, dtype=dtype)
    with torch.no_grad():
        direct = K.geglu_ln_direct_ok(x, w)
    if direct:
        g, beta, wp = torch.randn(Kd, device=device) * 0.2 + 1, torch.randn(Kd, device=device), K.pack_geglu_frag80(w)
        def run():
            with torch.no_grad():
                return K.geglu_ln_direct(x, g, beta, 1e-5, wp, b, N // 2)
        arm = "direct"
    else:
        w32, b32 = interleave_geglu(w, b)
        w8, b8 = interleave_geglu(w, b, 8)
        run = lambda: K.geglu_linear(x, w, b, w32, b32, w8, b8)
    for _ in range(3):
        run()
    if not direct:
        arm = K._choice.get(("geglu", M, N, Kd))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * M * N * Kd
    achieved = flops / (ms * 1e-3) / 1e12
    names = {"direct": "geglu_direct_kernel<640> (LayerNorm + GEGLU projection, A resident, weights in fragment order)", 0: "vendor library + geglu_kernel",
             3: "gemm_kernel<256x256,16 waves,GEGLU>", 13: "gemm8_kernel<256x256,8-phase,GEGLU>", 512: "gemm160p_kernel<160x320 persistent,GEGLU>"}
    out = {"bound": "mfma", "kernel": f"{names.get(arm, f'fmc_linear_bf16 arm {arm}')} [{M}x{N}x{Kd}]", "autotuned_arm": arm,
           "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
           "avg_launch_ms": round(ms, 4), "flops_per_launch": flops, "traffic_algorithmic": 2.0 * (M * Kd + N * Kd + M * N // 2)}
    out.update(recorded_counters("proj"))
    out["_match"] = ("name", "geglu_direct_kernel<640") if direct else None
    return out


def measure_proj_l0_roofline(device, dtype, iters=20):
    """The feed-forward input projection of the 40x64 level as the U-Net issues it since round 6: LayerNorm + GEGLU projection `[81920 tokens, 320] x
    [2560, 320]^T` gated to 1280 columns, the gate software-pipelined under the next chunk's MFMAs (`fmc_geglu_pipe_ln_bf16`, 160-row tiles), tile-major
    output for the second GEMM.  K = 320: ~15 gate instructions per 1280 matrix flops per output -- the VALU stream is as long as the MFMA stream."""
    from synfmc_amd import hip_ops as K
    M, Kd, N = 2 * FRAMES * (HEIGHT // 8) * (WIDTH // 8), WIDTHS[0], 8 * WIDTHS[0]
    x = torch.randn(M, Kd, device=device, dtype=dtype)
    w = torch.randn(N, Kd, device=device, dtype=dtype) * Kd ** -0.5
    b = torch.randn(N, device=device, dtype=dtype)
    g, beta = torch.randn(Kd, device=device) * 0.2 + 1, torch.randn(Kd, device=device)
    with torch.no_grad():
        pipe = K.geglu_ln_pipe_ok(x, w)
        var = K.geglu_pipe_variant(M, Kd) if pipe else -1
        wp = K.pack_geglu_frag(w, 16 if var == 1 else 32) if pipe else K.pack_geglu_frag80(w)

        def run():
            with torch.no_grad():
                if pipe:
                    return K.geglu_ln_pipe(x, g, beta, 1e-5, wp, b, N // 2, blocked=True, variant=var)
                return K.geglu_ln_direct(x, g, beta, 1e-5, wp, b, N // 2, blocked=True)
        ms = _time_launch(run, iters)
    flops = 2.0 * M * N * Kd
    achieved = flops / (ms * 1e-3) / 1e12
    name = f"geglu_pipe_kernel<320, {'8 waves x 160 rows' if var == 1 else '4 waves x 80 rows'}>" if pipe else "geglu_direct_kernel<320>"
    out = {"bound": "mfma", "kernel": f"{name} (LayerNorm + GEGLU projection, A resident, gate pipelined under the MFMAs) [{M}x{N}x{Kd}]",
           "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "avg_launch_ms": round(ms, 4),
           "flops_per_launch": flops, "traffic_algorithmic": 2.0 * (M * Kd + N * Kd + M * N // 2), "traffic": None,
           "_match": ("name", "geglu_pipe_kernel<320" if pipe else "geglu_direct_kernel<320")}
    if pipe and var == 1:
        out.update(recorded_counters("proj_l0"))
    return out


def measure_linear_l0_roofline(device, dtype, iters=30):
    """One of the K = 320 token projections of the 40x64 level as the U-Net issues them 25 times per step (`to_out`, `proj_in` / `proj_out` of the spatial
    and temporal transformers: M = 81920 tokens, 320 -> 320, bias + residual), through the autotuned front-end.  16.8 GF on 157 MB: HBM bound
    (x + residual read, out written, 0.2 MB of weight)."""
    from synfmc_amd import hip_ops as K
    M, N, Kd = 2 * FRAMES * (HEIGHT // 8) * (WIDTH // 8), WIDTHS[0], WIDTHS[0]
    x = torch.randn(M, Kd, device=device, dtype=dtype)
    r = torch.randn(M, N, device=device, dtype=dtype)
    w = torch.randn(N, Kd, device=device, dtype=dtype) * Kd ** -0.5
    b = torch.randn(N, device=device, dtype=dtype)
    with torch.no_grad():
        ms = _time_launch(lambda: K.linear(x, w, b, residual=r), iters)
    arm = K._choice.get(("lin", M, N, Kd, True, 1, 0))
    nbytes = 2.0 * (M * Kd + 2 * M * N + N * Kd)
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": f"fmc_linear_bf16 arm {arm} (K = 320 projection + bias + residual of the 40x64 level) [{M}x{N}x{Kd}]", "autotuned_arm": arm,
            "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4), "avg_launch_ms": round(ms, 5), "bytes_per_launch": nbytes,
            "flops_per_launch": 2.0 * M * N * Kd, "traffic": None,
            "_match": ("own_linear", lambda sh, M=M, N=N, Kd=Kd: sh[:3] == (M, N, Kd) and sh[4] == 1 and str(sh[3]).startswith("t"))}


def measure_ff2_roofline(device, dtype, iters=20):
    """The feed-forward's OUTPUT projection of the 40x64 level as the U-Net issues it behind the LayerNorm + GEGLU kernel.  Round 6: folded with the
    transformer's proj_out into ONE product (`hip_ops.ff_tail`): `[g | h] [Wp W2 | Wp]^T + b' + x`, `[81920, 1600] x [320, 1600]^T`, the 210-MB g in the
    tile-major layout the GEGLU kernel leaves, h / x / out row-major: 84 GF on 367 MB, arithmetic intensity 229 flop / B, below the chip's ridge (312):
    HBM bound.  (`FMC_FF_TAIL=0`: the un-folded `linear_from_blocked`, 67 GF on 315 MB, followed by a separate proj_out launch of 157 MB.)"""
    from synfmc_amd import hip_ops as K
    M, N, Kd = 2 * FRAMES * (HEIGHT // 8) * (WIDTH // 8), WIDTHS[0], 4 * WIDTHS[0]
    xb = torch.randn(M, Kd, device=device, dtype=dtype)
    h = torch.randn(M, N, device=device, dtype=dtype)
    r = torch.randn(M, N, device=device, dtype=dtype)
    w = torch.randn(N, Kd, device=device, dtype=dtype) * Kd ** -0.5
    b = torch.randn(N, device=device, dtype=dtype)
    if not K.geglu_direct_blocked_ok(h, w, r):
        return None
    wp = torch.randn(N, N, device=device, dtype=dtype) * N ** -0.5
    with torch.no_grad():
        tail = K.ff_tail_ok(h, w, wp, r)
        if tail:
            wc, bc = K.fold_ff_tail(w, b, wp, b)
            ms = _time_launch(lambda: K.ff_tail(xb, h, wc, bc, r, gn_hw=(HEIGHT // 8) * (WIDTH // 8)), iters)
            Kd2 = Kd + N
            nbytes, flops = 2.0 * (M * Kd + 3 * M * N + N * Kd2), 2.0 * M * N * Kd2
            what, tag = "feed-forward output projection folded with proj_out + bias + residual + GroupNorm partials, two-segment reduction", "ff-tail"
        else:
            ms = _time_launch(lambda: K.linear_from_blocked(xb, w, b, r), iters)
            Kd2 = Kd
            nbytes, flops = 2.0 * (M * Kd + 2 * M * N + N * Kd), 2.0 * M * N * Kd
            what, tag = "feed-forward output projection + bias + residual", "from-blocked"
    gbs = nbytes / (ms * 1e-3) / 1e9
    out = {"bound": "hbm", "kernel": f"gemm160p_kernel<blocked A> ({what}, tile-major intermediate) [{M}x{N}x{Kd2}]",
           "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4), "avg_launch_ms": round(ms, 5), "bytes_per_launch": nbytes,
           "flops_per_launch": flops, "mfma_frac_isolated": round(flops / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), "traffic": None,
           "_match": ("own_linear", lambda sh, M=M, N=N, Kd2=Kd2, tag=tag: sh[:4] == (M, N, Kd2, tag))}
    if tail:
        out.update(recorded_counters("ff2"))
    return out


def measure_vendor_roofline(device, dtype, call_log, iters=30):
    """The problem on which the step spends most of its hipBLASLt time (from the recorded call order of one eager step: the vendor-arm shape with the largest
    calls x flops -- at the time of writing `[5120, 1280] x [1280, 1280]^T + bias + residual`, the out-projections of the 10x16 level), through the same
    front-end (`hip_ops.linear` -> `fmc_vendor_linear_bf16` with the candidate the arm table holds).  MFMA bound."""
    from synfmc_amd import hip_ops as K
    tot = {}
    for fe, shape, fl in call_log or []:
        if fe == "vendor":
            tot[shape] = tot.get(shape, 0.0) + fl
    if tot:
        (M, N, Kd, has_b, has_r) = max(tot, key=tot.get)
    else:
        (M, N, Kd, has_b, has_r) = (2 * FRAMES * (HEIGHT // 32) * (WIDTH // 32), 3 * WIDTHS[2], WIDTHS[2], False, False)
    x = torch.randn(M, Kd, device=device, dtype=dtype)
    w = torch.randn(N, Kd, device=device, dtype=dtype) * Kd ** -0.5
    b = torch.randn(N, device=device, dtype=dtype) if has_b else None
    r = torch.randn(M, N, device=device, dtype=dtype) if has_r else None
    v0 = K.vendor_direct_calls["direct"]
    with torch.no_grad():                                                   # (the direct library call is the inference path's)
        ms = _time_launch(lambda: K.linear(x, w, b, residual=r), iters)
    vendor = K.vendor_direct_calls["direct"] > v0
    arm = K._choice.get(("lin", M, N, Kd, bool(has_b), int(bool(has_r)), 0))
    flops = 2.0 * M * N * Kd
    achieved = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": (f"hipBLASLt through fmc_vendor_linear_bf16 (candidate {K._choice.get(('valgo', M, N, Kd, Kd, N if has_r else 0, bool(has_b), bool(has_r)))})"
                                        if vendor else f"fmc_linear_bf16 arm {arm}") + f" [{M}x{N}x{Kd}{' + bias' if has_b else ''}{' + residual' if has_r else ''}]",
            "calls_per_step": sum(1 for fe, sh, _ in call_log or [] if fe == "vendor" and sh == (M, N, Kd, has_b, has_r)),
            "autotuned_arm": arm, "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
            "avg_launch_ms": round(ms, 4), "flops_per_launch": flops, "traffic_algorithmic": 2.0 * (M * Kd + N * Kd + M * N), "traffic": None,
            "_match": ("vendor", (M, N, Kd, has_b, has_r)) if vendor else ("own_linear", lambda sh, M=M, N=N, Kd=Kd: sh[:3] == (M, N, Kd))}


def measure_conv_halo4_roofline(device, dtype, iters=20):
    """The ResNet convolution of the 10x16 level (CFG batch 2 x 16 frames, 1280 -> 1280; `conv_halo4_kernel<16>`: 320 pixels of whole row blocks x 80
    channels, 256 tiles) through the front-end.  Algorithmic bytes x + w + out = 55.7 MB: 29.5 MB of it the filter, which every pixel tile streams."""
    from synfmc_amd import hip_ops as K
    n, h, w, ci, co = 2 * FRAMES, HEIGHT // 32, WIDTH // 32, WIDTHS[2], WIDTHS[2]
    x = torch.randn(n, h, w, ci, device=device, dtype=dtype).permute(0, 3, 1, 2)
    wt = (torch.randn(co, ci, 3, 3, device=device, dtype=dtype) * 0.02).contiguous(memory_format=torch.channels_last)
    log0, K.call_log = K.call_log, []
    ms = _time_launch(lambda: K.conv3x3(x, wt, None), iters)
    halo4 = any(fe == "conv_halo4" for fe, _, _ in K.call_log)
    K.call_log = log0
    flops = 2.0 * n * h * w * 9 * ci * co
    achieved = flops / (ms * 1e-3) / 1e12
    out = {"bound": "mfma", "kernel": ("conv_halo4_kernel<16> (input halo resident, 4 waves, software-pipelined)" if halo4 else "fmc_conv3x3_bf16 (autotuned arm)")
                                      + f" [{n}x{h}x{w}, {ci}->{co}]",
           "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "avg_launch_ms": round(ms, 4),
           "flops_per_launch": flops, "traffic_algorithmic": 2.0 * (n * h * w * (ci + co) + 9 * ci * co), "traffic": None,
           "_match": ("conv_halo4", (n, h, w, ci, co, False)) if halo4 else None}
    if halo4:
        out.update(recorded_counters("conv_halo4"))
    return out


def measure_temporal_block_l1_roofline(device, dtype, iters=20):
    """The same fused block at the 20x32 level (C = 640, 8 heads x 80; `temporal_block640.hip`: 80-row tiles, weights streamed in fragment order into
    registers), as the U-Net issues it: CFG batch 2 x 640 pixels x 16 frames = 20480 rows = 256 tiles, one per CU."""
    from synfmc_amd import hip_ops as K
    B, Fr, hw, C, H = 2, FRAMES, (HEIGHT // 16) * (WIDTH // 16), WIDTHS[1], 8
    if not (Fr == 16 and C == 640 and hw % 5 == 0):
        return None
    h = torch.randn(B, Fr, hw, C, device=device, dtype=dtype)
    pt = torch.randn(B, Fr, hw, C, device=device, dtype=dtype)
    g = torch.randn(C, device=device) * 0.2 + 1
    bpe = torch.randn(Fr, C, device=device)
    wq = torch.randn(3 * C, C, device=device, dtype=dtype) * C ** -0.5
    wo = torch.randn(C, C, device=device, dtype=dtype) * C ** -0.5
    wm = torch.randn(C, C, device=device, dtype=dtype) * C ** -0.5
    bo = torch.randn(C, device=device, dtype=dtype)
    wqp, wot, wmt = K.pack_temporal_qkv80(wq, H), K.pack_w_frag80(wo), K.pack_w_frag80(wm)
    run = lambda: K.temporal_block(h, g, bpe, 1e-5, wqp, wot, bo, (C // H) ** -0.5, w_merge_tm=wmt, pose_term=pt, merge_scale=1.0)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    M = B * Fr * hw
    flops = 2.0 * M * C * 5 * C + 4.0 * M * Fr * C
    achieved = flops / (ms * 1e-3) / 1e12
    out = {"bound": "mfma", "kernel": f"temporal_block640_kernel<merge> (fused LN + merge + qkv + attention + out-projection, bf16) [{B}x{Fr}x{hw}x{C}]",
           "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
           "avg_launch_ms": round(ms, 4), "flops_per_launch": flops, "traffic_algorithmic": 3.0 * M * C * 2,
           "replaces": "LayerNorm + merge GEMM + fused QKV GEMM + temporal_attn_kernel + out-projection GEMM (5 launches, ~160 us)"}
    out.update(recorded_counters("tblock640"))
    out["_match"] = ("name", "temporal_block640_kernel<true, false>")
    return out


def measure_temporal_block_roofline(device, dtype, iters=20):
    """The north-star's temporal-attention target ("MFMA utilisation on temporal attention"): one launch of the FUSED attention block of a
    40x64-level motion module exactly as the U-Net issues it (`fmc_temporal_block_bf16`: LayerNorm + pe -> Camera-Adapter merge + pose term ->
    q | k | v -> attention over the 16 frames -> out-projection + residual; CFG batch 2 x 2560 pixels x 16 frames = 81920 rows, C = 320).
    Algorithmic flops = 2 M C (C + 3 C + C) [merge, q | k | v, out] + 4 M F C [scores + PV]; HBM-side bytes h + pose term + out."""
    from synfmc_amd import hip_ops as K
    B, Fr, hw, C, H = 2, FRAMES, (HEIGHT // 8) * (WIDTH // 8), WIDTHS[0], 8
    if not (Fr == 16 and C == 320 and hw % 10 == 0):
        return None
    h = torch.randn(B, Fr, hw, C, device=device, dtype=dtype)
    pt = torch.randn(B, Fr, hw, C, device=device, dtype=dtype)
    g = torch.randn(C, device=device) * 0.2 + 1
    bpe = torch.randn(Fr, C, device=device)
    wq = torch.randn(3 * C, C, device=device, dtype=dtype) * C ** -0.5
    wo = torch.randn(C, C, device=device, dtype=dtype) * C ** -0.5
    wm = torch.randn(C, C, device=device, dtype=dtype) * C ** -0.5
    bo = torch.randn(C, device=device, dtype=dtype)
    wqp, wot, wmt = K.pack_temporal_qkv(wq, H), K._w_tilemajor(wo), K._w_tilemajor(wm)
    run = lambda: K.temporal_block(h, g, bpe, 1e-5, wqp, wot, bo, (C // H) ** -0.5, w_merge_tm=wmt, pose_term=pt, merge_scale=1.0)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    M = B * Fr * hw
    flops = 2.0 * M * C * 5 * C + 4.0 * M * Fr * C
    achieved = flops / (ms * 1e-3) / 1e12
    out = {"bound": "mfma", "kernel": f"temporal_block_kernel<merge> (fused LN + merge + qkv + attention + out-projection, bf16) [{B}x{Fr}x{hw}x{C}]",
           "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
           "avg_launch_ms": round(ms, 4), "flops_per_launch": flops, "traffic_algorithmic": 3.0 * M * C * 2,
           "replaces": "LayerNorm epilogue + merge GEMM + fused QKV GEMM + temporal_attn_kernel + out-projection GEMM (4 launches, ~790 MB of HBM traffic)"}
    out.update(recorded_counters("tblock"))
    out["_match"] = ("name", "temporal_block_kernel<true, false, false>")
    return out


def torch_op_sites(fn, path):
    """Diagnostic: device time of the torch (aten) ops of one eager call of `fn` by Python call site -> `path` (which lines still launch torch
    elementwise / copy / cat kernels)."""
    import collections
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        fn()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0.0, 0])
    for ev in prof.events():
        if ev.device_type.name != "CPU" or not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
            continue
        dev_us = sum(k.duration for k in ev.kernels) if ev.kernels else 0.0
        if dev_us <= 0:
            continue
        frames = [f for f in (ev.stack or []) if "synfmc_amd" in f or "bench.py" in f][:3]
        where = " <- ".join(fr.split("/")[-1] for fr in frames)
        if not where:                                    # backward: no Python stack -- name the autograd node the op ran under
            root = ev
            while root.cpu_parent is not None:
                root = root.cpu_parent
            where = root.name
        key = (ev.name, where)
        agg[key][0] += dev_us
        agg[key][1] += 1
    # the profiler does not always hand back Python stacks (graph-runner calls under no_grad): a dispatch-mode pass names the call sites of the
    # large elementwise / copy ops with their element counts
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    sites = collections.defaultdict(lambda: [0, 0])

    class Sites(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = func.__name__ if hasattr(func, "__name__") else str(func)
            if isinstance(out, torch.Tensor) and out.is_cuda and out.numel() >= (1 << 20) and any(k in str(func) for k in ("add", "cat", "mul", "copy", "silu", "clone", "contiguous")):
                st = [f"{os.path.basename(fr.filename)}:{fr.lineno}" for fr in traceback.extract_stack() if "synfmc_amd" in fr.filename and "hip_ops" not in fr.filename][-3:]
                key = (str(func), " <- ".join(reversed(st)))
                sites[key][0] += 1
                sites[key][1] += out.numel()
            return out
    with Sites():
        fn()
        torch.cuda.synchronize()
    with open(path, "w") as f:
        for (name, where), (us, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
            f.write(f"{us / 1e3:8.3f} ms {n:5d}x {name:28s} {where}\n")
        f.write("\ncall sites of the large elementwise / copy ops (dispatch mode): count, output elements\n")
        for (name, where), (n, el) in sorted(sites.items(), key=lambda kv: -kv[1][1])[:60]:
            f.write(f"{n:5d}x {el / 1e6:9.1f} M  {name:24s} {where}\n")


def in_step_trace(args, cfg, nsteps=4):
    """Per-launch kernel durations INSIDE the denoising step: this same command (fewer steps, no oracle, no roofline loops) run once more as a
    child process under `rocprofv3 --kernel-trace`, cut to the window of its last `nsteps` steps (between `cfg_ddim_kernel` dispatches).  The
    isolated roofline loops above flatter or punish a kernel by up to 20 % (warm weights, DVFS in a back-to-back loop): `frac` of every roofline
    object is computed from THIS window when it is available.  Returns ([per step: [(kernel name, grid x, workgroup x, us)] in launch order], note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="fmc_in_step_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "t", "--", sys.executable, os.path.abspath(__file__),
           "--steps", str(nsteps + 2), "--warmup", "2", "--trace-child", "--config", cfg, "--guidance", str(args.guidance), "--dtype", args.dtype]
    if args.no_cfg_shared:
        cmd.append("--no-cfg-shared")
    if args.fp8_temporal:
        cmd.append("--fp8-temporal")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, timeout=900, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        files = glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None, f"trace child failed (rc {r.returncode}): {r.stderr.decode(errors='replace')[-300:]}"
        rows = list(csv.DictReader(open(files[0])))
    except Exception as e:                                  # the trace is evidence, never a reason to lose the bench line
        return None, f"trace child: {type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [int(r["End_Timestamp"]) for r in rows if "cfg_ddim" in r["Kernel_Name"]]
    if len(ends) < nsteps + 1:
        return None, f"only {len(ends)} step boundaries in the trace"
    steps = []
    for a, b in zip(ends[-nsteps - 1:-1], ends[-nsteps:]):
        steps.append([(r["Kernel_Name"], int(r.get("Grid_Size_X", 0) or 0), int(r.get("Workgroup_Size_X", 0) or 0),
                       (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                      for r in rows if int(r["Start_Timestamp"]) > a and int(r["End_Timestamp"]) <= b])
    return steps, f"rocprofv3 --kernel-trace of a child run of this command, last {nsteps} steps"


# kernel-name filters of the launch families `hip_ops._log_call` records (one kernel launch of the family per logged call, in program order)
_FAMILY_KERNELS = {
    "conv_halo": ("conv_halo_kernel<",),
    "conv_halo4": ("conv_halo4_kernel<",),
    "vendor": ("Cijk_",),
    "own_linear": ("gemm160p_kernel<", "gemm160_kernel<0", "gemm8_kernel<0", "gemm_kernel<0", "gemm_k320_kernel<", "gemm4_kernel"),
    "geglu_direct": ("geglu_direct_kernel<", "geglu_pipe_kernel<"),
    "fused_block": ("temporal_block_kernel<", "temporal_block640_kernel<"),
}


def family_durations(steps, call_log, family):
    """[(logged shape, [in-step us of that launch in every traced step])] for one launch family, by launch ORDER: the i-th launch of the family's
    kernels inside a step is the i-th call the eager step logged.  None when the counts differ (an arm that launches a second matching kernel)."""
    order = [shape for fe, shape, _ in call_log if fe == family]
    subs = _FAMILY_KERNELS[family]
    per_call = [[] for _ in order]
    for st in steps or []:
        ks = [d for name, gx, wx, d in st if any(x in name for x in subs)]
        if len(ks) != len(order):
            return None
        for i, d in enumerate(ks):
            per_call[i].append(d)
    return list(zip(order, per_call)) if steps else None


def apply_in_step(obj, steps, call_log):
    """Add `in_step_avg_ms` / `in_step_calls_per_step` to one roofline object and recompute `frac` from it (the isolated loop's figure stays as
    `frac_isolated`).  `_match`: ("name", substring[, grid x]) = every in-window launch of that kernel, or (family, shape | predicate) = the launches
    the recorded call order of one eager step (`call_log`) assigns to that shape within a launch family (`_FAMILY_KERNELS`)."""
    if obj is None:
        return None
    match = obj.pop("_match", None)
    if steps is None or match is None:
        obj["in_step_avg_ms"] = None
        return obj
    us = []
    if match[0] == "name":
        hits = [(gx, d) for st in steps for name, gx, wx, d in st if match[1] in name]
        if len(match) > 3 and match[2] == "grid":                # one shape of a kernel that serves several: its grid (threads in x)
            hits = [h for h in hits if h[0] == match[3]]
        if len(match) > 2 and match[2] == "maxgrid" and hits:    # the launches of the FULL batch (the CFG-shared prefix runs the same kernel on half of it)
            g = max(gx for gx, _ in hits)
            hits = [h for h in hits if h[0] == g]
        us = [d for _, d in hits]
    else:
        fam = family_durations(steps, call_log, match[0])
        want = match[1] if callable(match[1]) else (lambda shape, w=tuple(match[1]): shape == w)
        if fam is None:
            obj["in_step_avg_ms"] = None
            obj["in_step_note"] = f"launch count of the {match[0]} kernels in the traced step differs from the recorded call order: no per-shape assignment"
            return obj
        for shape, ds in fam:
            if want(shape):
                us += ds
    if not us:
        obj["in_step_avg_ms"] = None
        obj["in_step_note"] = "the step does not launch this kernel on this shape (fused away or another arm)"
        return obj
    ms = sum(us) / len(us) / 1e3
    obj["in_step_avg_ms"] = round(ms, 4)
    obj["in_step_calls_per_step"] = round(len(us) / len(steps), 2)
    work = obj["flops_per_launch"] if obj["unit"] == "TFLOP/s" else obj["bytes_per_launch"]     # (HBM-bound objects may quote their flops too)
    scale = 1e12 if obj["unit"] == "TFLOP/s" else 1e9
    obj["frac_isolated"] = obj["frac"]
    obj["achieved_isolated"] = obj["achieved"]
    obj["achieved"] = round(work / (ms * 1e-3) / scale, 2)
    obj["frac"] = round(obj["achieved"] / obj["peak"], 4)
    return obj


def step_kernel_families(steps, call_log):
    """ms per step by kernel family + the halo convolutions' aggregate rate (their shapes from the recorded call order), from the in-step window."""
    if steps is None:
        return None
    fam = (("conv3x3 (conv_halo_kernel)", ("conv_halo_kernel", "conv_halo4_kernel", "conv_halo4_finish")), ("conv3x3 / linear (gemm*_kernel, sk_finish)", ("gemm", "sk_finish", "splitk")),
           ("vendor GEMM (hipBLASLt Cijk_*)", ("Cijk_",)), ("geglu_direct / geglu_pipe", ("geglu_direct", "geglu_pipe")), ("spatial attention", ("sa40d", "spatial_attn", "sa_small160", "sa_big80")),
           ("temporal block / attention", ("temporal_block", "temporal_attn")), ("text cross-attention block", ("xattn",)),
           ("groupnorm", ("gn_",)), ("layernorm", ("layernorm",)), ("torch elementwise / copy / cat", ("at::", "elementwise", "CatArray")))
    tot = {k: 0.0 for k, _ in fam}
    tot["other"] = 0.0
    calls = 0
    for st in steps:
        for name, gx, wx, d in st:
            calls += 1
            for k, subs in fam:
                if any(x in name for x in subs):
                    tot[k] += d
                    break
            else:
                tot["other"] += d
    n = len(steps)
    out = {"ms_per_step": {k: round(v / n / 1e3, 3) for k, v in tot.items()}, "launches_per_step": round(calls / n, 1)}
    # the token GEMMs of the step: every projection launch -- own gemm* kernels, the vendor arm, the resident-operand GEGLU
    # projections -- and, apart, the fused attention blocks (their flops include the projections they absorbed)
    def family_total(fams, subs):
        fl_ = sum(f for fe, _, f in call_log if fe in fams)
        us_ = sum(d for st in steps for name, gx, wx, d in st if any(x in name for x in subs)) / n
        if not fl_ or not us_:
            return None
        return {"launches_per_step": sum(1 for fe, _, _ in call_log if fe in fams), "tflop_per_step": round(fl_ / 1e12, 3), "ms_per_step": round(us_ / 1e3, 3),
                "achieved": round(fl_ / (us_ * 1e-6) / 1e12, 1), "frac": round(fl_ / (us_ * 1e-6) / 1e12 / PEAK_BF16_TFLOPS, 4), "unit": "TFLOP/s"}
    lin_subs = _FAMILY_KERNELS["own_linear"] + _FAMILY_KERNELS["vendor"] + _FAMILY_KERNELS["geglu_direct"] + ("splitk_reduce",)
    out["linear_all_launches"] = family_total(("own_linear", "vendor", "geglu_direct"), lin_subs)
    out["linear_own_gemm_launches"] = family_total(("own_linear",), _FAMILY_KERNELS["own_linear"] + ("splitk_reduce",))
    out["linear_vendor_launches"] = family_total(("vendor",), _FAMILY_KERNELS["vendor"])
    out["geglu_direct_launches"] = family_total(("geglu_direct",), _FAMILY_KERNELS["geglu_direct"])
    out["fused_blocks_all_launches"] = family_total(("fused_block",), _FAMILY_KERNELS["fused_block"])
    fl = sum(f for fe, _, f in call_log if fe in ("conv_halo", "conv_halo4"))
    ms = tot["conv3x3 (conv_halo_kernel)"] / n / 1e3
    if fl and ms:
        out["conv_halo_all_launches"] = {"launches_per_step": sum(1 for fe, _, _ in call_log if fe in ("conv_halo", "conv_halo4")), "tflop_per_step": round(fl / 1e12, 3),
                                         "ms_per_step": round(ms, 3), "achieved": round(fl / (ms * 1e-3) / 1e12, 1),
                                         "frac": round(fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), "unit": "TFLOP/s"}
    return out


def unet_flops(batch, h, w, executed=False, config="obj"):
    """Analytic forward FLOPs from a meta-device trace of the oracle.  `executed=False`: the reference graph (LoRA as
    separate `up(down(x))` GEMMs, text K/V projected once per FRAME).  `executed=True`: what the product launches --
    LoRA merged into the projection weights, text K/V projected once per CLIP outside the step (not counted)."""
    from torch.utils.flop_counter import FlopCounterMode
    from oracle import fmc_modules as OM
    from synfmc_amd import configs as CM
    with torch.device("meta"):
        u = OM.UNet3DConditionModelCamObjCond(**CM.unet_kwargs(WIDTHS, CROSS_DIM))
        u.set_all_attn_processor(**CM.processor_kwargs(WIDTHS, lora=not executed, temporal=config != "lora"))
        if config == "obj":
            OM.patch_down_blocks_for_omc(u)
        x, text = torch.empty(batch, 4, FRAMES, h, w), torch.empty(batch, 77, CROSS_DIM)
        feats = [torch.empty(batch, c, FRAMES, h // s, w // s) for c, s in zip(WIDTHS, (1, 2, 4, 8))]
        with FlopCounterMode(display=False) as fc:
            u(x, torch.empty(batch, dtype=torch.long), text, pose_embedding_features=None if config == "lora" else feats,
              traj_features=feats if config == "obj" else None)
    total = float(fc.get_total_flops())
    if executed:      # 16 cross-attention layers: K and V of the 77 text tokens are projected ONCE PER CLIP (outside the step, `Attention.text_kv`):
        per_level = {0: 5, 1: 5, 2: 5, 3: 1}                   # the reference graph's `frames` copies per step all go (down 2 + up 3 per level, mid at level 3)
        total -= sum(n * 2 * 2.0 * batch * FRAMES * 77 * CROSS_DIM * WIDTHS[l] for l, n in per_level.items())
    return total


def measure_temporal_roofline(device, dtype, iters=50):
    """Level-0 temporal attention exactly as the U-Net launches it: q/k/v slices of one fused [2, 16, 2560, 960] projection
    (CFG batch 2 clips x 2560 pixels x 8 heads x 16 frames).  HBM-bound: arithmetic intensity F/2 = 8 flop/B."""
    from synfmc_amd import hip_ops as K
    B, P, H, D = 2, (HEIGHT // 8) * (WIDTH // 8), 8, WIDTHS[0] // 8
    C = H * D
    qkv = torch.randn(B, FRAMES, P, 3 * C, device=device, dtype=dtype)
    for _ in range(3):
        K.self_attention_qkv(qkv, H, D ** -0.5, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        K.self_attention_qkv(qkv, H, D ** -0.5, True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = 4.0 * B * P * FRAMES * C * 2                      # q, k, v read + o written, bf16
    gbs = nbytes / (ms * 1e-3) / 1e9
    out = {"bound": "hbm", "kernel": f"temporal_attn_kernel<bf16,d={D}> [2x{P} pixels x {H} heads, F={FRAMES}]",
           "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
           "avg_launch_ms": round(ms, 5), "bytes_per_launch": nbytes}
    out.update(recorded_counters("temporal"))
    out["_match"] = None               # (at this level the step runs the attention inside temporal_block_kernel: `roofline_temporal_block`)
    out["in_step_note"] = "not launched by the step at this level (fused into temporal_block_kernel); the isolated loop is the only measurement"
    return out


def measure_temporal_fp8_roofline(device, iters=50):
    """Level-0 temporal attention of a TRAINING step on the fp8 path (BASELINE configs[4]): e4m3 q | k | v of one clip
    ([1, F, pixels, 3 x 320] bytes out of the QKV projection's epilogue) -> bf16 o.  HBM-bound: 3 + 2 bytes per (token, channel)."""
    from synfmc_amd import hip_ops as K
    P, H, D = (HEIGHT // 8) * (WIDTH // 8), 8, WIDTHS[0] // 8
    C = H * D
    qkv8 = (torch.randn(1, FRAMES, P, 3 * C, device=device) * 0.5).to(torch.float8_e4m3fn)
    sc = torch.ones(3, dtype=torch.float32, device=device)
    for _ in range(3):
        K._temporal_fp8_raw(qkv8, sc, H, D ** -0.5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        K._temporal_fp8_raw(qkv8, sc, H, D ** -0.5)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = 5.0 * P * FRAMES * C
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": f"temporal_attn_fp8_kernel<d={D}> [1x{P} pixels x {H} heads, F={FRAMES}; e4m3 q|k|v, bf16 o]",
            "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
            "avg_launch_ms": round(ms, 5), "bytes_per_launch": nbytes, "traffic": None,
            "traffic_note": "PMC traffic of this launch: profiles/r02_temporal_pmc.md (136.5 MB at F=16, CFG batch 2); not re-collected for this shape"}


def oracle_step(unet, enc, ada, clip, text2, latents, t, want_config1=True, config="obj", full=False):
    """ONE step of the metric's configuration on the host cores through the oracle (fp32 restatement of the reference),
    with the benchmarked model's own (bf16-rounded) weights and the benchmark's own inputs: CFG-batch-2 16x320x512 U-Net +
    CMC + OMC forward.  Returns (eps fp32 `[2,4,F,h,w]`, cpu_baseline dict).  The camera encoder and the OMC adapter run
    once per clip outside the step, as in the pipeline; their time is reported but not part of the step."""
    from einops import rearrange
    from oracle import conditioning as OC
    from oracle import fmc_modules as OM
    from synfmc_amd import configs as CM
    # oneDNN / OpenMP scaling collapses far below the 256 hardware threads of the GPU node (same 16x128x192 sample: 2.3 s on
    # 8 threads, 1.6 s on 16, 2.6 s on 32, 5.6 s on 64, 427 s on 256), so the port runs on a fixed, stated number of threads
    cores = min(os.cpu_count() or 1, int(os.environ.get("FMC_CPU_BASELINE_THREADS", "16")))
    torch.set_num_threads(cores)
    t_build = time.time()
    with torch.device("meta"):
        ou = OM.UNet3DConditionModelCamObjCond(**CM.unet_kwargs(WIDTHS, CROSS_DIM))
        ou.set_all_attn_processor(**CM.processor_kwargs(WIDTHS, temporal=config != "lora"))
        oe = OM.CameraPoseEncoder(**CM.encoder_kwargs(WIDTHS, max(16, FRAMES))) if enc is not None else None
        oa = OM.Adapter(**CM.adapter_kwargs(WIDTHS)) if ada is not None else None
    for o, p in ((ou, unet), (oe, enc), (oa, ada)):
        if o is None:
            continue
        o.to_empty(device="cpu")
        o.load_state_dict({k: v.detach().float().cpu() for k, v in p.state_dict().items()}, strict=True)
        o.eval()
    if config == "obj":
        OM.patch_down_blocks_for_omc(ou)
    t_build = time.time() - t_build
    with torch.no_grad():
        t0 = time.time()
        pose2 = traj2 = None
        if oe is not None:
            pose_emb = rearrange(OC.to_plucker_embedding(clip["c2w"], clip["K"], (HEIGHT, WIDTH)), "b f c h w -> b c f h w")
            pose = [rearrange(x, "(b f) c h w -> b c f h w", b=1) for x in oe(pose_emb)]
            pose2 = [torch.cat([x, x]) for x in pose]                                 # pipeline_animation_cm_om.py:668-669
        if oa is not None:
            traj = OC.get_traj_features(clip["infos"], clip["masks"], oa)
            traj2 = [torch.cat([torch.zeros_like(x), x]) for x in traj]               # :671-676
        t_cond = time.time() - t0
        x2 = torch.cat([latents, latents]).to(torch.bfloat16).float()                 # the values the GPU path is fed
        # warm-up: one forward of the SAME oracle on a 16x128x128 slice of the benchmark's own inputs (~2 s: oneDNN primitive creation, thread pool,
        # allocator) before the timed step -- SURVEY 8d asks for a warm-up; three timed repeats of a 45-s step do not fit the default run
        t0 = time.time()
        hw = 16
        ou(x2[..., :hw, :hw].contiguous(), torch.tensor(int(t)), text2.float().cpu(),
           pose_embedding_features=None if pose2 is None else [p[..., :max(1, hw >> i), :max(1, hw >> i)].contiguous() for i, p in enumerate(pose2)],
           traj_features=None if traj2 is None else [p[..., :max(1, hw >> i), :max(1, hw >> i)].contiguous() for i, p in enumerate(traj2)])
        t_warm = time.time() - t0
        t_all = []
        for _ in range(4 if full else 1):      # `full` (--cpu-baseline-full): SURVEY 8d's protocol -- 1 full-size warm-up + 3 timed steps
            t0 = time.time()
            eps = ou(x2, torch.tensor(int(t)), text2.float().cpu(), pose_embedding_features=pose2, traj_features=traj2).sample
            t_all.append(time.time() - t0)
        t_step = sum(t_all[1:]) / 3 if full else t_all[0]
        t_c1 = None
        if want_config1:           # BASELINE configs[0]: 1x16x256x256 fp32, base U-Net, plain processors, no adapters
            ou.set_attn_processor(OM.AttnProcessor())
            ou.set_mm_attn_processor(OM.AttnProcessor())
            g = torch.Generator().manual_seed(7)
            x1, txt1 = torch.randn(1, 4, 16, 32, 32, generator=g), torch.randn(1, 77, CROSS_DIM, generator=g)
            t0 = time.time()
            ou(x1, torch.tensor(500), txt1)
            t_c1 = time.time() - t0
    del ou, oe, oa
    base = {"value": round(1.0 / t_step, 6), "unit": "denoising steps/s", "cores": cores, "host_cpus": os.cpu_count(),
            "kind": "port",
            "sample": f"{'the mean of three real steps' if full else 'ONE real step'} of the benchmarked configuration ({config}), not extrapolated: oracle (fp32 restatement; the reference "
                      f"needs diffusers) U-Net{'' if config == 'lora' else '+CMC'}{'+OMC' if config == 'obj' else ''} forward at CFG "
                      f"batch 2 on the 16x320x512 clip = {t_step:.2f} s on "
                      f"{cores} threads (thread policy: min(cpu_count, 16), oneDNN scaling collapses beyond; "
                      + (f"mean of 3 timed steps {[round(v, 2) for v in t_all[1:]]} s after one full-size warm-up step of {t_all[0]:.2f} s (SURVEY 8d protocol, "
                         f"--cpu-baseline-full)" if full else
                         f"ONE timed step (n = 1: three repeats of a 45-s step do not fit the default run; --cpu-baseline-full runs 1 warm-up + 3 timed) after a "
                         f"{t_warm:.1f}-s warm-up forward of the same oracle on a 128x128 slice of the same inputs") + f"); conditioning once per clip (Pluecker + camera encoder + OMC adapter) {t_cond:.2f} s; "
                      f"weights copy {t_build:.1f} s"
                      + (f"; BASELINE configs[0] (1x16x256x256 fp32 base U-Net, no adapters) forward {t_c1:.2f} s = "
                         f"{1.0 / t_c1:.4f} steps/s" if t_c1 else "")}
    return eps, base


def fp32_parity_mode_line(args, cfg, device, models, clip, text2, eps_ref, t_par, steps=5, warmup=2):
    """The SAME workload in the fp32-storage parity mode (north_star's 1e-3 rel-inf is met by this mode, so the driver-run line
    carries its parity figure and its rate): the benchmarked model's own (bf16-rounded) weights in fp32 tensors, every GEMM / conv / attention as
    split-bf16 x3 MFMA on the same hand-written kernels, one step against the oracle's output for the identical inputs, then
    `steps` timed graph replays of the denoising step."""
    from synfmc_amd import hip_ops as K
    from synfmc_amd.models.pose_adaptor import features_to_video
    from synfmc_amd.pipelines.pipeline_animation_cm_om import _GraphedUNet
    from synfmc_amd.schedulers import DDIMScheduler
    from synfmc_amd.util import stack_object_inputs
    f32 = torch.float32
    t0 = time.time()
    m32 = build_models(device, f32, cfg)
    for a, b in zip(m32, models):
        if a is not None:
            a.load_state_dict({k: v.float() for k, v in b.state_dict().items()}, strict=True)
    unet, enc, ada = m32
    text = text2.float()
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
    sched.set_timesteps(50, device=device)
    with torch.no_grad():
        pose_feats = traj_feats = None
        if enc is not None:
            poses, masks = stack_object_inputs(clip["infos"], clip["masks"], device)
            emb = K.plucker(clip["K"].to(device), clip["c2w"].to(device), HEIGHT, WIDTH, "unshuffle8", f32)
            pose_feats = [torch.cat([x, x], 0).contiguous(memory_format=torch.channels_last_3d) for x in features_to_video(enc.forward_unshuffled(emb, 1), 1)]
            if ada is not None:
                feats, m = K.omc_rasterize(poses, masks, "unshuffle8", f32)
                traj_feats = [t.contiguous(memory_format=torch.channels_last_3d) for t in features_to_video(ada(feats, m), 1)]
        latents = clip["latents"].to(device).float().contiguous()
        runner = _GraphedUNet(unet, (2,) + tuple(latents.shape[1:]), text, pose_feats, traj_feats, f32, cfg_shared_input=not args.no_cfg_shared)
        unet.prepare_text_conditioning(runner.text)
        runner.capture()
        x_par = torch.cat([latents, latents]).to(torch.bfloat16).float()              # the values the bf16 path and the oracle were fed
        eps = runner(x_par, t_par).float().cpu()
        parity = float((eps - eps_ref).abs().max() / eps_ref.abs().max()) if eps_ref is not None else None
        ts = sched._timesteps_host

        def step(i):
            nonlocal latents
            t = ts[i % len(ts)]
            latents = sched.step_cfg(runner(torch.cat([latents, latents]), t), t, latents, args.guidance, True)
        elapsed = timed_region(step, steps, warmup, 1, torch.cuda.synchronize)
        assert torch.isfinite(latents).all(), "non-finite latents (fp32 parity mode)"
    out = {"dtype": "fp32 storage, split-bf16 x3 MFMA products (parity mode, not the metric's dtype)", "steps": steps, "warmup": warmup,
           "steps_per_s": round(steps / elapsed, 3), "ms_per_step": round(elapsed / steps * 1e3, 2),
           "parity_rel_inf": parity, "parity_gate": 1e-3,
           "note": f"same workload, weights, inputs and oracle output as the bf16 line; north_star's tolerance (1e-3 rel-inf vs the CPU reference) is this "
                   f"mode's gate; built + captured in {time.time() - t0:.1f} s"}
    if parity is not None and not (parity < 1e-3):
        raise SystemExit(f"bench.py: fp32 parity mode FAILED its gate -- rel-inf {parity:.3e} >= 1e-3 against the CPU oracle")
    del runner, unet, enc, ada, m32
    torch.cuda.empty_cache()
    return out


# --------------------------------------------------------------------------------------------------------------
# launch plumbing shared by every mode
# --------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def ensure_ranks(args) -> None:
    """`--gpus N` must mean N ranks.  Without a launcher (no WORLD_SIZE in the environment) and N > 1 this process
    re-executes itself under `python -m torch.distributed.run --nproc-per-node N` and exits with its status; under a
    launcher a WORLD_SIZE that disagrees with --gpus is an error, not a warning."""
    if "WORLD_SIZE" not in os.environ:
        if args.gpus <= 1:
            return
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        log(f"bench.py: starting {args.gpus} ranks: {' '.join(cmd)}")
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")))
    world = int(os.environ["WORLD_SIZE"])
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")


def init_ranks(backend: str, device):
    """(rank, local_rank, world) with the world size VERIFIED by a collective: every rank contributes 1, the sum is what
    `n_gpus` reports."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        kw = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)
        one = torch.ones(1, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(one)
        if int(one.item()) != world:
            raise SystemExit(f"bench.py: all-reduce of ones returned {int(one.item())}, expected {world}")
        world = int(one.item())
    return rank, local_rank, world


def timed_region(step_fn, steps: int, warmup: int, world: int, sync) -> float:
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides; MAX over ranks (seconds)."""
    for i in range(warmup):
        step_fn(i)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(warmup + i)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        el = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if torch.cuda.is_available() and dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    return elapsed


def dry_run_main(args):
    """Launcher / timing / JSON plumbing on CPU with the gloo backend and a stub step (no model, no kernels): what the 2-rank
    CPU test of this entry point runs.  Never a measurement: the line says `"dry_run": true`."""
    ensure_ranks(args)
    rank, _, world = init_ranks("gloo", torch.device("cpu"))
    x = torch.randn(64, 64, generator=torch.Generator().manual_seed(rank))

    def step(i):
        nonlocal x
        x = torch.tanh(x @ x.t() / 64.0)

    elapsed = timed_region(step, args.steps, args.warmup, world, lambda: None)
    if rank == 0:
        print(json.dumps({"metric": "DRY RUN (launcher / timing plumbing only)", "dry_run": True, "value": round(world * args.steps / elapsed, 4),
                          "unit": "stub steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "stub", "parallelism": f"dp{world}", "backend": "gloo"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def train_main(args):
    """Secondary measurement: stage-3 (OMC) training steps/s, one clip per GPU, weak scaling, gradients averaged
    over ranks by `synfmc_amd.training.GradAllReducer` (RCCL).  Reference loop: train_cam_obj_ctrl.py:782-943.

    HIP graphs: one rank = the whole step (forward, backward, clip, AdamW) in ONE graph.  Several ranks = graph A (forward +
    backward into the flat gradient buckets) | RCCL all-reduce of the buckets (eager, a handful of large messages) | graph B
    (clip + AdamW + zeroing): the step stays graph-replayed, only the exchange is issued from the host.  `--train-graph`
    forces either form (or none) on any world size."""
    ensure_ranks(args)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    rank, local_rank, world = init_ranks("nccl", device)
    global FRAMES, HEIGHT, WIDTH
    FRAMES, HEIGHT, WIDTH = (int(v) for v in args.clip.lower().split("x"))
    dtype = torch.bfloat16
    from synfmc_amd.models.pose_obj_adaptor import CamObjPoseAdaptor
    from synfmc_amd.schedulers import DDIMScheduler
    from synfmc_amd.training import (GradAllReducer, biased_timesteps, broadcast_parameters, optimizer_update,
                                     stage3_forward_backward)
    from synfmc_amd.util import stack_object_inputs
    from synfmc_amd import hip_ops as K
    from synfmc_amd.models.pose_adaptor import features_to_video
    from synfmc_amd import configs as TC
    unet, enc, ada = build_models(device, dtype)
    ada = ada.float().requires_grad_(True)                  # fp32 master weights for the trainable Adapter
    if args.fp8_temporal:
        from synfmc_amd.models.motion_module import enable_fp8_temporal_attention
        enable_fp8_temporal_attention(unet)
        enable_fp8_temporal_attention(enc)
    broadcast_parameters(ada)
    clip, _ = synthetic_inputs(rank, device)
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear",
                          steps_offset=1, clip_sample=False)
    mode = args.train_graph or ("one" if world == 1 else "split")
    if args.no_graph:
        mode = "none"
    # overlap (all-reduce launched from inside backward) cannot live inside a captured graph: graphed modes reduce after it
    reducer = GradAllReducer(ada.parameters(), overlap=(mode == "none"),
                             compress_dtype=torch.bfloat16 if args.grad_compress == "bf16" else None)
    wrapper = CamObjPoseAdaptor(unet, enc)
    poses, masks = stack_object_inputs(clip["infos"], clip["masks"], device)
    c2w, Kin = clip["c2w"].to(device), clip["K"].to(device)
    latents, text = clip["latents"].to(device).to(dtype), clip["text"].to(device).to(dtype)
    obj_masks = TC.union_masks(clip).to(device)
    gen = torch.Generator(device=device).manual_seed(77 + rank)

    noise = torch.empty(latents.shape, device=device, dtype=dtype)    # static inputs of the (graphed) step
    t = torch.zeros(1, device=device, dtype=torch.long)

    def draw():
        noise.copy_(torch.randn(latents.shape, device=device, dtype=dtype, generator=gen))
        t.copy_(biased_timesteps(1, 1000, 700, 0.8, device, gen))

    def fwd_bwd():
        emb = K.plucker(Kin, c2w, HEIGHT, WIDTH, "bcfhw", dtype)      # on device, every step (reference: CPU + H2D)

        def traj_fn():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                feats, m = K.omc_rasterize(poses, masks, "unshuffle8", dtype)
                return features_to_video(ada(feats, m), 1)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return stage3_forward_backward(wrapper, sched, latents, noise, t, text, emb, traj_fn, obj_masks)

    # discovery step (eager): autotune, MIOpen find, and the reducer learns which parameters are never used (Adapter level 3)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        draw()
        fwd_bwd()
        reducer.finish()
        reducer.zero_grad()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    trainable = reducer.parameters()                         # the used subset; the rest keeps grad = None (as under DDP)
    opt = torch.optim.AdamW(trainable, lr=1e-6, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, capturable=mode != "none")

    def update():
        optimizer_update(trainable, opt, reducer, 1.0)

    if mode == "none":
        def step(_i=0):
            draw()
            loss = fwd_bwd()
            reducer.finish()
            update()
            return loss
    else:
        with torch.cuda.stream(side):                        # eager warm-up of the optimizer state before capture
            for _ in range(2):
                draw()
                fwd_bwd()
                reducer.finish()
                update()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        draw()
        if mode == "one":
            # one rank: the whole step is one graph.  Several ranks (opt-in, `--train-graph one`): the bucket all-reduces are CAPTURED
            # with the step -- RCCL collectives are capturable stream operations, c10d registers the communicator with the capture --
            # so the exchange replays from the graph behind the Adapter's backward without a host round trip.  The default on
            # several ranks stays `split` until this form has run on an 8-GPU node (tests/test_gpu_multi.py tries both).
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_loss = fwd_bwd()
                reducer.finish()
                update()

            def step(_i=0):
                draw()
                graph.replay()
                return static_loss
        else:
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga):
                static_loss = fwd_bwd()
            reducer.finish()                                 # (eager; completes the step the capture recorded)
            with torch.cuda.graph(gb, pool=ga.pool()):
                update()

            def step(_i=0):
                draw()
                ga.replay()
                for b in reducer.buckets:                    # what zero_grad() re-arms on the host side
                    b["launched"] = False
                reducer.finish()                             # RCCL all-reduce of the flat buckets + 1/world scaling
                gb.replay()
                return static_loss

    loss = None

    if getattr(args, "torch_profile", None):                 # diagnostic: which call sites launch the torch elementwise / copy kernels
        draw()

        def one():
            fwd_bwd()
            reducer.finish()
        torch_op_sites(one, args.torch_profile)
        reducer.zero_grad()

    def run(i):
        nonlocal loss
        loss = step(i)

    elapsed = timed_region(run, args.steps, args.warmup, world, torch.cuda.synchronize)
    assert torch.isfinite(loss).all()
    # every rank holds the same averaged gradients / parameters: the first trained tensor must agree bit for bit across ranks
    params_equal = None
    if world > 1:
        probe = trainable[0].detach().float().reshape(-1)[:4096].clone()
        lo, hi = probe.clone(), probe.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        params_equal = bool(torch.equal(lo, hi))
        assert params_equal, "trained parameters differ across ranks after the all-reduced steps"
    if rank == 0:
        with torch.no_grad():
            roof = measure_attention_roofline(device, dtype, clips=1)
            roof_t = measure_temporal_fp8_roofline(device) if args.fp8_temporal else None
        roof_bwd = measure_attention_bwd_roofline(device, dtype)
        for o in (roof, roof_t):
            if o is not None:
                o.pop("_match", None)
        print(json.dumps({
            "metric": f"OMC-stage training steps/sec (secondary), {FRAMES}x{HEIGHT}x{WIDTH} bf16"
                      f"{' + fp8 temporal attention' if args.fp8_temporal else ''}, frozen U-Net + trainable Adapter",
            "value": round(world * args.steps / elapsed, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16 (e4m3 q|k|v in the temporal attention)" if args.fp8_temporal else "bf16",
            "data": "synthetic",
            "config": {"workload": "stage-3 (configs/obj.yaml) training step, 1 clip per GPU, AdamW, clip-norm 1.0, "
                                   "bucketed RCCL all-reduce of the USED Adapter gradients (level 3 never receives one)",
                       "hip_graph": mode, "parallelism": f"dp{world}", "allreduce_bytes": reducer.allreduce_bytes(),
                       "fp8_temporal_attention": args.fp8_temporal, "grad_compress": args.grad_compress, "unused_params": sum(p.numel() for p in reducer.unused),
                       "trained_params": sum(p.numel() for p in trainable), "baseline_config": "train32" if args.config == "train32" else None,
                       "parameters_bit_equal_across_ranks": params_equal},
            "last_loss": float(loss), "roofline": roof, "roofline_temporal": roof_t, "roofline_attention_bwd": roof_bwd, "cpu_baseline": None,
            "cpu_baseline_note": "reported with the metric's configuration only (python bench.py)"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle step (no parity_rel_inf, no cpu_baseline)")
    ap.add_argument("--guidance", type=float, default=8.0)
    ap.add_argument("--autotune-log", default=None, help="write the per-shape GEMM/conv arm timings to this file")
    ap.add_argument("--mode", default="infer", choices=["infer", "train"],
                    help="infer = the headline metric (denoising steps/s); train = OMC-stage optimisation steps/s "
                         "(secondary: forward + activation backward through the frozen U-Net + Adapter backward + RCCL "
                         "gradient all-reduce + AdamW), 16x256x384 like configs/obj.yaml")
    ap.add_argument("--clip", default="16x256x384", help="train mode: frames x height x width of the clip "
                    "(16x256x384 = configs/obj.yaml; 32x512x512 = BASELINE configs[4] in bf16)")
    ap.add_argument("--train-graph", default=None, choices=["one", "split", "none"],
                    help="train mode: one HIP graph for the whole step (1 rank) | graph + all-reduce + graph | eager")
    ap.add_argument("--grad-compress", default="none", choices=["none", "bf16"], help="train mode: gradient buckets on the wire")
    ap.add_argument("--fp8-temporal", action="store_true",
                    help="temporal attention on the fp8 path (e4m3 q|k|v from the QKV epilogue, fp8 MFMA): BASELINE configs[4]")
    ap.add_argument("--torch-profile", default=None, help="diagnostic: write the device time of torch ops by call site (one eager step) to this file")
    ap.add_argument("--config", default="obj", choices=["obj", "cam", "lora", "train32"],
                    help="which BASELINE.json configuration: obj = configs[3], the metric's workload (default); cam = configs[2] (CMC only); "
                         "lora = configs[1] (Domain-LoRA only, 50-step DDIM loop); train32 = configs[4] (32x512x512 stage-3 training step, "
                         "fp8 temporal attention; = --mode train --clip 32x512x512 --fp8-temporal)")
    ap.add_argument("--no-cfg-shared", action="store_true", help="A/B: compute the CFG batch's identical prefix (conv_in, first ResNet block, "
                    "first self-attention) for both halves instead of once")
    ap.add_argument("--no-loop50", action="store_true", help="skip the 50-step DDIM loop through the configuration's own pipeline (ddim_50_step_loop_*)")
    ap.add_argument("--no-fp32-line", action="store_true", help="skip the fp32-storage parity-mode sub-line (fp32_parity_mode)")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="cpu_baseline as SURVEY 8d words it: 1 full-size warm-up step + 3 timed steps of the "
                                                                     "oracle (adds ~3 min); default: ONE timed step after a small-slice warm-up")
    ap.add_argument("--no-in-step", action="store_true", help="skip the in-step kernel trace (a child run of this command under rocprofv3 --kernel-trace)")
    ap.add_argument("--trace-child", action="store_true", help="internal: the child run of the in-step trace (no oracle, no roofline loops, no trace)")
    ap.add_argument("--dry-run", action="store_true", help="CPU / gloo stub of the launcher + timing + JSON plumbing (tests)")
    args = ap.parse_args()
    if args.dry_run:
        return dry_run_main(args)
    if args.config == "train32":
        args.mode, args.clip, args.fp8_temporal = "train", "32x512x512", True
    if args.mode == "train":
        return train_main(args)

    ensure_ranks(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs MI355X GPUs: the HIP path has no CPU fallback")
    cfg = args.config
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    rank, local_rank, world = init_ranks("nccl", device)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    from synfmc_amd.models.pose_adaptor import features_to_video
    from synfmc_amd.pipelines.pipeline_animation_cm_om import _GraphedUNet
    from synfmc_amd.schedulers import DDIMScheduler
    from synfmc_amd.util import stack_object_inputs
    from synfmc_amd import hip_ops as K

    t_build = time.time()
    unet, enc, ada = build_models(device, dtype, cfg)
    if args.fp8_temporal:
        from synfmc_amd.models.motion_module import enable_fp8_temporal_attention
        enable_fp8_temporal_attention(unet)
        if enc is not None:
            enable_fp8_temporal_attention(enc)
    clip, text2 = synthetic_inputs(rank, device)
    text2 = text2.to(dtype)
    torch.cuda.synchronize()
    log(f"[rank {rank}] config {cfg}: models built in {time.time() - t_build:.1f} s")

    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear",
                          steps_offset=1, clip_sample=False)
    sched.set_timesteps(50, device=device)

    # ---- once per clip: Pluecker rays + camera encoder, OMC rasteriser + adapter (outside the loop, as in the reference)
    pose_feats = traj_feats = None
    cond_ms = 0.0
    if enc is not None:
        poses, masks = stack_object_inputs(clip["infos"], clip["masks"], device)
        c2w, Kin = clip["c2w"].to(device), clip["K"].to(device)
        torch.cuda.synchronize()
        with torch.no_grad():
            def conditioning():
                emb = K.plucker(Kin, c2w, HEIGHT, WIDTH, "unshuffle8", dtype)
                pf = features_to_video(enc.forward_unshuffled(emb, 1), 1)
                tf = None
                if ada is not None:
                    feats, m = K.omc_rasterize(poses, masks, "unshuffle8", dtype)
                    tf = features_to_video(ada(feats, m), 1)
                return pf, tf
            conditioning()
            torch.cuda.synchronize()
            t0 = time.time()
            pose_feats, traj_feats = conditioning()
            torch.cuda.synchronize()
            cond_ms = (time.time() - t0) * 1e3
        pose_feats = [torch.cat([x, x], 0).contiguous(memory_format=torch.channels_last_3d) for x in pose_feats]
        if traj_feats is not None:
            traj_feats = [t.contiguous(memory_format=torch.channels_last_3d) for t in traj_feats]

    latents = clip["latents"].to(device).float().contiguous()
    x_shape = (2,) + tuple(latents.shape[1:])
    parity, cpu, parity_tol = None, None, (4e-2 if dtype == torch.bfloat16 else 1e-3)
    # every step feeds the U-Net cat([latents, latents]) (as the reference pipeline does): the prefix the two halves share runs once
    cfg_shared = not args.no_cfg_shared
    with torch.no_grad():
        runner = _GraphedUNet(unet, x_shape, text2, pose_feats, traj_feats, dtype, cfg_shared_input=cfg_shared)
        # once per clip as well: k | v of the text tokens for the 16 cross-attention layers + their MFMA-fragment packs --
        # the reference re-projects them in every step (fmc/models/attention_processor.py:58-59); here no step launches them
        from synfmc_amd.models import layers as L_
        text_in = text2 if args.no_graph else runner.text
        n_text_layers = unet.prepare_text_conditioning(text_in)
        torch.cuda.synchronize()
        t0 = time.time()
        for m in unet.modules():
            if m.__dict__.get("_text_kv") is not None:
                m.refresh_text_kv()
        torch.cuda.synchronize()
        text_ms = (time.time() - t0) * 1e3
        cond_ms += text_ms
        if args.no_graph:
            def unet_step(x, t):
                kw = {}
                if pose_feats is not None:
                    kw["pose_embedding_features"] = pose_feats
                    if cfg == "obj":
                        kw["traj_features"] = traj_feats
                return unet(x, torch.tensor(int(t), device=device), encoder_hidden_states=text2, **({'cfg_shared_input': True} if cfg_shared else {}), **kw).sample
        else:
            runner.capture()
            unet_step = runner

        # ---- parity GATE: one step of THIS model on THESE inputs against the CPU oracle, before anything is timed.  A kernel path that
        # does not reproduce the oracle never prints a throughput line; an oracle that throws takes the run down with it.  Runs at
        # N = 1 (where `cpu_baseline` is reported); at N > 1 no rank waits 45 s in a barrier for rank 0's CPU forward -- every rank's
        # kernels are the ones gated at N = 1, and the ranks check their outputs for finiteness.
        if world == 1 and not args.no_cpu_baseline and not args.trace_child:
            t_par = 801
            eps_gpu = unet_step(torch.cat([latents, latents]).to(dtype), t_par).float().cpu()
            eps_ref, cpu = oracle_step(unet, enc, ada, clip, text2, clip["latents"].float(), t_par, config=cfg, full=args.cpu_baseline_full)
            parity = float((eps_gpu - eps_ref).abs().max() / eps_ref.abs().max())
            log(f"[rank 0] parity of the benchmarked model vs the CPU oracle: rel-inf {parity:.3e} ({args.dtype}, gate {parity_tol:g}); "
                f"oracle step {1.0 / cpu['value']:.1f} s")
            if not (parity < parity_tol):
                raise SystemExit(f"bench.py: PARITY GATE FAILED -- rel-inf {parity:.3e} >= {parity_tol:g} against the CPU oracle "
                                 f"({args.dtype}, config {cfg}); nothing was timed")

        def denoise_step(i):
            nonlocal latents
            t = ts[i % len(ts)]
            x = torch.cat([latents, latents]).to(dtype)
            eps = unet_step(x, t)
            latents = sched.step_cfg(eps, t, latents, args.guidance, True)

        ts = sched._timesteps_host
        text_kv_before = dict(L_.text_kv_calls)
        elapsed = timed_region(denoise_step, args.steps, args.warmup, world, torch.cuda.synchronize)
        text_kv_in_loop = L_.text_kv_calls["computed"] - text_kv_before["computed"]
        assert text_kv_in_loop == 0, f"{text_kv_in_loop} text k | v projections ran inside the timed steps"
        assert torch.isfinite(latents).all(), "non-finite latents"

        call_log, dispatch = [], None
        if rank == 0 and not args.trace_child:
            # one EAGER step of exactly what the graph replays (right behind the timed steps: the 50-step pipeline below installs its own per-clip
            # caches on the modules): which front-end calls went to an own kernel / the vendor arm / fell through as ineligible, and the order and
            # shapes of every GEMM-shaped launch (to assign the in-step trace's launches to shapes)
            K.call_log = []
            d0 = {k: dict(v) for k, v in K.dispatch_calls.items()}
            calls0 = {k: c for k, _, _, c in K.autotune_report()}
            if args.no_graph:
                unet_step(torch.cat([latents, latents]).to(dtype), ts[0])
            else:
                runner._call()
            torch.cuda.synchronize()
            call_log, K.call_log = K.call_log, None
            if os.environ.get("FMC_BENCH_SHAPES"):
                # per-shape view of the step's autotuned front-end calls (GEMM / conv arms): calls per step, chosen arm, the tuner's own isolated time for it and
                # for the vendor arm -- where the step's GEMM time sits, shape by shape (a side file, not part of the JSON line)
                rows = []
                for k, arm, ms_, c in K.autotune_report():
                    n = c - calls0.get(k, 0)
                    if n > 0:
                        rows.append({"shape": list(k), "calls_per_step": n, "arm": arm, "ms_arm": ms_.get(arm), "ms_vendor": ms_.get(0),
                                     "ms_per_step": None if ms_.get(arm) is None else round(n * ms_[arm], 4)})
                rows.sort(key=lambda r: -(r["ms_per_step"] or 0))
                with open(os.environ["FMC_BENCH_SHAPES"], "w") as f:
                    for r in rows:
                        f.write(json.dumps(r) + "\n")
            dispatch = {k: {a: K.dispatch_calls[k][a] - d0[k][a] for a in v} for k, v in K.dispatch_calls.items()}

        loop50_s = None
        if rank == 0 and not args.trace_child and not args.no_loop50:
            # the metric as the reference runs it: the configuration's own pipeline end to end -- `CameraObjCtrlPipeline.__call__`
            # (fmc/pipelines/pipeline_animation_cm_om.py:570-738: camera encoder once, then 50 DDIM steps at CFG batch 2 with the OMC features) for obj /
            # cam, `AnimationPipeline`'s plain loop for lora -- on this clip, latents out (VAE / CLIP are outside the path), graphs warm from a first call
            from synfmc_amd.pipelines.pipeline_animation_cm_om import AnimationPipeline, CameraObjCtrlPipeline
            kw = dict(prompt=None, video_length=FRAMES, height=HEIGHT, width=WIDTH, num_inference_steps=50, guidance_scale=args.guidance,
                      prompt_embeds=text2, latents=clip["latents"].to(device), output_type="latent")
            if cfg == "lora":
                pipe = AnimationPipeline(None, None, None, unet, sched)
            else:
                pipe = CameraObjCtrlPipeline(None, None, None, unet, sched, enc)
                kw.update(pose_embedding=K.plucker(Kin, c2w, HEIGHT, WIDTH, "unshuffle8", dtype), pose_embedding_unshuffled=True)
                if cfg == "obj":
                    feats, m = K.omc_rasterize(poses, masks, "unshuffle8", dtype)
                    kw["traj_features"] = features_to_video(ada(feats, m), 1)
            pipe(**kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out50 = pipe(**kw)
            torch.cuda.synchronize()
            loop50_s = time.perf_counter() - t0
            assert torch.isfinite(torch.as_tensor(out50.videos)).all()

    fp32_line = None
    if rank == 0 and world == 1 and dtype == torch.bfloat16 and not args.trace_child and not args.no_fp32_line and not args.fp8_temporal:
        fp32_line = fp32_parity_mode_line(args, cfg, device, (unet, enc, ada), clip, text2, eps_ref if parity is not None else None, 801)
        log(f"[rank 0] fp32 parity mode: {fp32_line['steps_per_s']} steps/s, rel-inf {fp32_line['parity_rel_inf']}")
    if rank == 0 and args.trace_child:
        print(json.dumps({"trace_child": True, "ms_per_step": round(elapsed / args.steps * 1e3, 3)}), flush=True)
        return
    if rank == 0:
        bf = dtype == torch.bfloat16
        if getattr(args, "torch_profile", None):
            with torch.no_grad():
                torch_op_sites((lambda: unet_step(torch.cat([latents, latents]).to(dtype), ts[0])) if args.no_graph else runner._call, args.torch_profile)
        K.save_autotune_table()                         # (the trace child reads the arm table this process tuned)
        roof = measure_attention_roofline(device, dtype) if bf else None
        roof_conv = measure_conv_roofline(device, dtype, 1) if bf else None
        roof_conv0 = measure_conv_roofline(device, dtype, 0) if bf else None
        roof_gn = measure_groupnorm_roofline(device, dtype) if bf else None
        roof_temp = measure_temporal_roofline(device, dtype) if bf else None
        roof_proj = measure_proj_roofline(device, dtype) if bf else None
        roof_tb = measure_temporal_block_roofline(device, dtype) if bf else None
        roof_tb1 = measure_temporal_block_l1_roofline(device, dtype) if bf else None
        roof_sa1 = measure_attention_level_roofline(device, dtype, 1) if bf else None
        roof_sa2 = measure_attention_level_roofline(device, dtype, 2) if bf else None
        roof_proj0 = measure_proj_l0_roofline(device, dtype) if bf else None
        roof_lin0 = measure_linear_l0_roofline(device, dtype) if bf else None
        roof_ff2 = measure_ff2_roofline(device, dtype) if bf else None
        roof_vendor = measure_vendor_roofline(device, dtype, call_log) if bf else None
        roof_halo4 = measure_conv_halo4_roofline(device, dtype) if bf else None
        steps_tr, tr_note = (None, "skipped (--no-in-step / N > 1 / fp32)") if (args.no_in_step or world > 1 or not bf) else in_step_trace(args, cfg)
        roofs = [roof, roof_conv, roof_conv0, roof_gn, roof_temp, roof_proj, roof_tb, roof_tb1, roof_sa1, roof_sa2, roof_proj0, roof_lin0, roof_ff2, roof_vendor, roof_halo4]
        for o in roofs:
            apply_in_step(o, steps_tr, call_log)
        families = step_kernel_families(steps_tr, call_log)
        f_ref = unet_flops(2, HEIGHT // 8, WIDTH // 8, config=cfg)
        f_exec = unet_flops(2, HEIGHT // 8, WIDTH // 8, executed=True, config=cfg)
        if cfg_shared:                                  # one clip's worth of conv_in, ResNet block 0, proj_in, QKV, self-attention, out-projection
            toks, c0 = FRAMES * (HEIGHT // 8) * (WIDTH // 8), WIDTHS[0]
            f_exec -= 2.0 * toks * (9 * 4 * c0 + 2 * 9 * c0 * c0 + c0 * c0 + 3 * c0 * c0 + c0 * c0) + 4.0 * FRAMES * ((HEIGHT // 8) * (WIDTH // 8)) ** 2 * c0
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": CONFIGS[cfg]["metric"],
            "value": round(world * args.steps / elapsed, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": CONFIGS[cfg]["workload"], "baseline_config": cfg,
                       "frames": FRAMES, "height": HEIGHT, "width": WIDTH, "guidance_scale": args.guidance,
                       "hip_graph": not args.no_graph, "fp8_temporal_attention": args.fp8_temporal,
                       "cfg_shared_prefix": bool(cfg_shared),
                       "parallelism": f"dp{world} (independent clips, no collective)"},
            "parity_rel_inf": parity, "parity_gate": parity_tol if parity is not None else None,
            "parity_note": ("max|eps_gpu - eps_oracle| / max|eps_oracle| for one CFG-batch-2 step (t = 801) of the benchmarked "
                            "model: same weights (bf16-rounded), noise, text, camera poses, object masks; oracle = fp32 CPU; the run aborts "
                            "before timing when it is not below parity_gate") if parity is not None else
                           ("not evaluated in this run (" + ("N > 1: the gate runs at N = 1, where cpu_baseline is reported" if world > 1
                                                              else "--no-cpu-baseline") + ")"),
            "unet_tflop_per_step_executed": round(f_exec / 1e12, 3),
            "unet_tflop_per_step_reference_graph": round(f_ref / 1e12, 3),
            "executed_tflops_per_gpu": round(f_exec / 1e12 / (ms * 1e-3), 1),
            "conditioning_once_per_clip_ms": round(cond_ms, 2),
            "conditioning_note": (f"Pluecker + camera encoder + OMC rasteriser / adapter + text k | v of {n_text_layers} cross-attention layers "
                                  f"({round(text_ms, 2)} ms); none of it runs inside a step"),
            "roofline": roof, "roofline_conv": roof_conv, "roofline_conv_l0": roof_conv0, "roofline_groupnorm": roof_gn, "roofline_temporal": roof_temp,
            "roofline_temporal_block": roof_tb, "roofline_temporal_block_l1": roof_tb1, "roofline_proj": roof_proj,
            "roofline_attention_l1": roof_sa1, "roofline_attention_l2": roof_sa2,
            "roofline_proj_l0": roof_proj0, "roofline_linear_l0": roof_lin0, "roofline_ff2": roof_ff2, "roofline_vendor_gemm": roof_vendor, "roofline_conv_halo4": roof_halo4,
            "in_step_source": tr_note, "in_step_kernel_families": families,
            "autotune": {"shapes_from_this_builds_cache": K.autotune_sources["cache"], "shapes_from_tracked_default_table": K.autotune_sources["defaults"],
                         "shapes_tuned_in_this_run": max(0, len(K._choice) - K.autotune_sources["cache"] - K.autotune_sources["defaults"]),
                         "note": "arm per GEMM / conv shape: synfmc_amd/autotune_default_mi355x.json (tracked) unless this build already has a cache"},
            "step_dispatch": {"note": "front-end calls of one step: own kernel / autotuner chose the vendor arm / shape outside the own kernels "
                                      "(fell through to the vendor library)", **dispatch,
                              "halo_convs_per_step": sum(1 for fe, _, _ in call_log if fe == "conv_halo"),
                              "halo4_convs_per_step": sum(1 for fe, _, _ in call_log if fe == "conv_halo4")},
            "cpu_baseline": cpu,
        }
        out["fp32_parity_mode"] = fp32_line
        if loop50_s is not None:
            out["ddim_50_step_loop_note"] = ("the configuration's own pipeline end to end on this clip (camera encoder once, 50 DDIM steps at CFG batch 2, latents "
                                             "out), second call: graphs warm")
            out["ddim_50_step_loop_s"] = round(loop50_s, 3)
            out["ddim_50_step_loop_steps_per_s"] = round(50.0 / loop50_s, 3)
        print(json.dumps(out), flush=True)
        if args.autotune_log:
            with open(args.autotune_log, "w") as f:
                for key, use, times, calls in sorted(K.autotune_report(), key=lambda r: -r[3] * min(r[2].values() or [0])):
                    f.write(f"{key} -> arm {use}  calls {calls}  ms {times}\n")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
