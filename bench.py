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
the timed region is bracketed by barrier + synchronize and the MAX over ranks is used.

The JSON line also carries
  roofline     -- the dominant hand-written kernel (level-0 spatial self-attention, S=2560, d=40, 256 (batch,head)
                  pairs): algorithmic flops 4*B*H*S^2*d divided by its average launch duration measured here with
                  events on the launch stream; peak = 2.5 PFLOP/s dense bf16 MFMA (MI355X_MICROARCH.md);
  cpu_baseline -- the oracle (fp32 PyTorch restatement; the reference itself needs diffusers, not installable)
                  timed on this node's host cores on a bounded sample and scaled by analytic FLOPs.
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
    """Seeded random init on the module's device (N(0, std); norm gains around 1).  Zero-initialised layers of the
    reference (qkv_merge, zero convs, LoRA up, proj_out) get values too so the conditioning paths do real work."""
    g = torch.Generator(device=next(module.parameters()).device).manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            p.normal_(0.0, std, generator=g)
            if p.ndim == 1 and "norm" in name and name.endswith("weight"):
                p.add_(1.0)


def build_models(device, dtype):
    from synfmc_amd.adapter import Adapter
    from synfmc_amd.models.pose_adaptor import CameraPoseEncoder
    from synfmc_amd.models.unet import UNet3DConditionModelCamObjCond
    from synfmc_amd.modified_modules import patch_unet_for_omc
    from tests import common_models as CM
    with torch.device(device):
        unet = UNet3DConditionModelCamObjCond(**CM.unet_kwargs(WIDTHS, CROSS_DIM))
        unet.set_all_attn_processor(**CM.processor_kwargs(WIDTHS))
        enc = CameraPoseEncoder(**CM.encoder_kwargs(WIDTHS, max(16, FRAMES)))
        ada = Adapter(**CM.adapter_kwargs(WIDTHS))
    patch_unet_for_omc(unet)
    for i, m in enumerate((unet, enc, ada)):
        fast_init_(m, 1234 + i)
        m.to(dtype=dtype).eval().requires_grad_(False)
    # buffers created under the device context are fine; re-make the PE tables in fp32 precision then cast
    return unet, enc, ada


def synthetic_inputs(rank, device):
    from tests import common_models as CM
    clip = CM.synthetic_clip(B=1, Fr=FRAMES, H=HEIGHT, W=WIDTH, n_obj=3, cross_dim=CROSS_DIM, seed=1234 + rank)
    g = torch.Generator().manual_seed(99 + rank)
    uncond = torch.randn(1, 77, CROSS_DIM, generator=g)
    return clip, torch.cat([uncond, clip["text"]]).to(device)


def measure_attention_roofline(device, dtype, iters=20):
    """Level-0 spatial self-attention exactly as the U-Net launches it: q/k/v are slices of one fused [32, 2560, 960]
    projection (CFG batch 2 x 16 frames, 8 heads x 40)."""
    from synfmc_amd import hip_ops as K
    B, S, H, D = 2 * FRAMES, (HEIGHT // 8) * (WIDTH // 8), 8, WIDTHS[0] // 8
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
    return {"bound": "mfma", "kernel": "spatial_attn_kernel<bf16,d=40,self> [B*H=256,S=2560]", "achieved": round(achieved, 2),
            "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
            "avg_launch_ms": round(ms, 4), "flops_per_launch": flops,
            # HBM-side bytes per launch of this exact shape, from rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE, the
            # gfx950 correction of MI355X_MICROARCH.md); recorded, not re-measured here: profiles/r01_attn_pmc.md
            "traffic": 259.6e6, "traffic_algorithmic": 4.0 * B * S * H * D * 2}


def measure_conv_roofline(device, dtype, iters=20):
    """The GEMM / conv kernel is where most of the step goes (63 %): one of its launches exactly as the U-Net issues it
    (level-1 ResNet conv, CFG batch 2 x 16 frames of 20x32, 640 -> 640, 256x256 tile), for the record next to the
    attention roofline the north-star asks for."""
    from synfmc_amd import hip_ops as K
    n, h, w, ci, co = 2 * FRAMES, HEIGHT // 16, WIDTH // 16, WIDTHS[1], WIDTHS[1]
    x = torch.randn(n, h, w, ci, device=device, dtype=dtype)
    wt = (torch.randn(co, ci, 3, 3, device=device, dtype=dtype) * 0.02).contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        K.conv3x3_bf16(x, wt, None, None, None, tile=3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        K.conv3x3_bf16(x, wt, None, None, None, tile=3)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * n * h * w * 9 * ci * co
    achieved = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": f"gemm_kernel<conv3x3,256x256> [{n}x{h}x{w}, {ci}->{co}]", "achieved": round(achieved, 2),
            "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
            "avg_launch_ms": round(ms, 4), "flops_per_launch": flops}


def unet_flops(batch, h, w):
    """Analytic forward FLOPs of the reference graph (un-merged LoRA) from a meta-device trace of the oracle."""
    from torch.utils.flop_counter import FlopCounterMode
    from oracle import fmc_modules as OM
    from tests import common_models as CM
    with torch.device("meta"):
        u = OM.UNet3DConditionModelCamObjCond(**CM.unet_kwargs(WIDTHS, CROSS_DIM))
        u.set_all_attn_processor(**CM.processor_kwargs(WIDTHS))
        OM.patch_down_blocks_for_omc(u)
        x, text = torch.empty(batch, 4, FRAMES, h, w), torch.empty(batch, 77, CROSS_DIM)
        feats = [torch.empty(batch, c, FRAMES, h // s, w // s) for c, s in zip(WIDTHS, (1, 2, 4, 8))]
        with FlopCounterMode(display=False) as fc:
            u(x, torch.empty(batch, dtype=torch.long), text, pose_embedding_features=feats, traj_features=feats)
    return float(fc.get_total_flops())


def cpu_baseline(budget_s=25.0):
    """Oracle U-Net (+CMC+OMC injection), full width, fp32, on the host cores; bounded sample = 16x128x192 clip,
    batch 1 (no CFG); scaled to the metric's unit by analytic FLOPs (the oracle is the 'port')."""
    from oracle import fmc_modules as OM
    from tests import common_models as CM
    # oneDNN/OpenMP scaling collapses far below the 256 hardware threads of the GPU node (measured: 427 s with 256, 5.6 s with 64, 2.6 s with 32, 1.6 s with 16, 2.3 s with 8
    # threads for the same sample), so the port is timed on a fixed, stated number of cores
    cores = min(os.cpu_count() or 1, int(os.environ.get("FMC_CPU_BASELINE_THREADS", "16")))
    torch.set_num_threads(cores)
    sh, sw = 128 // 8, 192 // 8
    with torch.device("meta"):
        u = OM.UNet3DConditionModelCamObjCond(**CM.unet_kwargs(WIDTHS, CROSS_DIM))
        u.set_all_attn_processor(**CM.processor_kwargs(WIDTHS))
    u = u.to_empty(device="cpu").eval()
    OM.patch_down_blocks_for_omc(u)
    block = torch.randn(1 << 20, generator=torch.Generator().manual_seed(0)) * 0.02
    with torch.no_grad():
        for p in u.parameters():                      # cheap deterministic fill (values do not affect timing)
            flat = p.view(-1)
            n = flat.numel()
            reps = (n + block.numel() - 1) // block.numel()
            flat.copy_(block.repeat(reps)[:n])
        for m in u.modules():
            if isinstance(m, OM.PositionalEncoding):
                m.pe.copy_(OM.PositionalEncoding(m.pe.shape[-1], max_len=m.pe.shape[1]).pe)
    g = torch.Generator().manual_seed(1)
    x, text = torch.randn(1, 4, FRAMES, sh, sw, generator=g), torch.randn(1, 77, CROSS_DIM, generator=g)
    feats = [torch.randn(1, c, FRAMES, sh // s, sw // s, generator=g) * 0.1 for c, s in zip(WIDTHS, (1, 2, 4, 8))]
    times = []
    with torch.no_grad():
        t0 = time.time()
        u(x, torch.tensor([801]), text, pose_embedding_features=feats, traj_features=feats)       # warm-up
        warm = time.time() - t0
        while len(times) < 3 and (sum(times) + warm) < budget_s:
            t0 = time.time()
            u(x, torch.tensor([801]), text, pose_embedding_features=feats, traj_features=feats)
            times.append(time.time() - t0)
    t_sample = min(times) if times else warm
    f_sample, f_step = unet_flops(1, sh, sw), unet_flops(2, HEIGHT // 8, WIDTH // 8)
    return {"value": round(1.0 / (t_sample * f_step / f_sample), 6), "unit": "denoising steps/s", "cores": cores,
            "kind": "port",
            "sample": f"oracle fp32 full-width U-Net+CMC+OMC forward, 16x128x192 clip, batch 1: {t_sample:.2f} s "
                      f"({f_sample / 1e12:.2f} TFLOP); scaled by FLOPs to a CFG-batch-2 16x320x512 step "
                      f"({f_step / 1e12:.2f} TFLOP)"}


def train_main(args):
    """Secondary measurement: stage-3 (OMC) training steps/s, one clip per GPU, weak scaling, gradients averaged
    over ranks by `synfmc_amd.training.GradAllReducer` (RCCL).  Reference loop: train_cam_obj_ctrl.py:782-943."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)
    global FRAMES, HEIGHT, WIDTH
    FRAMES, HEIGHT, WIDTH = (int(v) for v in args.clip.lower().split("x"))
    dtype = torch.bfloat16
    from synfmc_amd.models.pose_obj_adaptor import CamObjPoseAdaptor
    from synfmc_amd.schedulers import DDIMScheduler
    from synfmc_amd.training import GradAllReducer, biased_timesteps, broadcast_parameters, stage3_training_step
    from synfmc_amd.util import stack_object_inputs
    from synfmc_amd import hip_ops as K
    from synfmc_amd.models.pose_adaptor import features_to_video
    from tests import training_common as TC
    unet, enc, ada = build_models(device, dtype)
    ada = ada.float().requires_grad_(True)                  # fp32 master weights for the trainable Adapter
    broadcast_parameters(ada)
    clip, _ = synthetic_inputs(rank, device)
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear",
                          steps_offset=1, clip_sample=False)
    use_graph = world == 1 and not args.no_graph           # one rank: the whole step (fwd, bwd, clip, AdamW) is one HIP graph
    opt = torch.optim.AdamW([p for p in ada.parameters()], lr=1e-6, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8,
                            capturable=use_graph)
    reducer = GradAllReducer(ada.parameters())
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

    def body():
        emb = K.plucker(Kin, c2w, HEIGHT, WIDTH, "bcfhw", dtype)      # on device, every step (reference: CPU + H2D)

        def traj_fn():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                feats, m = K.omc_rasterize(poses, masks, "unshuffle8", dtype)
                return features_to_video(ada(feats, m), 1)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return stage3_training_step(wrapper, ada, sched, opt, reducer, latents, noise, t, text, emb, traj_fn, obj_masks)

    if use_graph:
        # eager warm-up on a side stream (autotune, MIOpen find, optimizer state), then capture the step once: eagerly the
        # step is launch-bound (34 ms of kernels in 77 ms of wall clock on one GPU)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                draw()
                body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        draw()
        with torch.cuda.graph(graph):
            static_loss = body()

        def step():
            draw()
            graph.replay()
            return static_loss
    else:
        def step():
            draw()
            return body()

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    assert torch.isfinite(loss).all()
    if rank == 0:
        print(json.dumps({
            "metric": f"OMC-stage training steps/sec (secondary), {FRAMES}x{HEIGHT}x{WIDTH} bf16, frozen U-Net + trainable Adapter",
            "value": round(world * args.steps / elapsed, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "stage-3 (configs/obj.yaml) training step, 1 clip per GPU, AdamW, clip-norm 1.0, "
                                   "bucketed RCCL all-reduce of 152.5M fp32 Adapter gradients",
                       "hip_graph": use_graph, "parallelism": f"dp{world}", "allreduce_bytes": sum(b["flat"].numel() * 4 for b in reducer.buckets)},
            "last_loss": float(loss)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--guidance", type=float, default=8.0)
    ap.add_argument("--autotune-log", default=None, help="write the per-shape GEMM/conv arm timings to this file")
    ap.add_argument("--mode", default="infer", choices=["infer", "train"],
                    help="infer = the headline metric (denoising steps/s); train = OMC-stage optimisation steps/s "
                         "(secondary: forward + activation backward through the frozen U-Net + Adapter backward + RCCL "
                         "gradient all-reduce + AdamW), 16x256x384 like configs/obj.yaml")
    ap.add_argument("--clip", default="16x256x384", help="train mode: frames x height x width of the clip "
                    "(16x256x384 = configs/obj.yaml; 32x512x512 = BASELINE configs[4] in bf16)")
    args = ap.parse_args()
    if args.mode == "train":
        return train_main(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs MI355X GPUs: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)
    if world != args.gpus and rank == 0:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    from synfmc_amd.data.dataset import to_plucker_embedding
    from synfmc_amd.models.pose_adaptor import features_to_video
    from synfmc_amd.pipelines.pipeline_animation_cm_om import _GraphedUNet
    from synfmc_amd.schedulers import DDIMScheduler
    from synfmc_amd.util import stack_object_inputs
    from synfmc_amd import hip_ops as K

    t_build = time.time()
    unet, enc, ada = build_models(device, dtype)
    clip, text2 = synthetic_inputs(rank, device)
    text2 = text2.to(dtype)
    torch.cuda.synchronize()
    log(f"[rank {rank}] models built in {time.time() - t_build:.1f} s")

    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear",
                          steps_offset=1, clip_sample=False)
    sched.set_timesteps(50, device=device)

    # ---- once per clip: Pluecker rays + camera encoder, OMC rasteriser + adapter (outside the loop, as in the reference)
    poses, masks = stack_object_inputs(clip["infos"], clip["masks"], device)
    c2w, Kin = clip["c2w"].to(device), clip["K"].to(device)
    torch.cuda.synchronize()
    with torch.no_grad():
        def conditioning():
            emb = K.plucker(Kin, c2w, HEIGHT, WIDTH, "unshuffle8", dtype)
            pf = features_to_video(enc.forward_unshuffled(emb, 1), 1)
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
    traj_feats = [t.contiguous(memory_format=torch.channels_last_3d) for t in traj_feats]

    latents = clip["latents"].to(device).float().contiguous()
    x_shape = (2,) + tuple(latents.shape[1:])
    with torch.no_grad():
        runner = _GraphedUNet(unet, x_shape, text2, pose_feats, traj_feats, dtype)
        if args.no_graph:
            def unet_step(x, t):
                return unet(x, torch.tensor(int(t), device=device), encoder_hidden_states=text2,
                            pose_embedding_features=pose_feats, traj_features=traj_feats).sample
        else:
            runner.capture()
            unet_step = runner

        def denoise_step(lat, t):
            x = torch.cat([lat, lat]).to(dtype)
            eps = unet_step(x, t)
            return sched.step_cfg(eps, t, lat, args.guidance, True)

        ts = sched._timesteps_host
        for i in range(args.warmup):
            latents = denoise_step(latents, ts[i % len(ts)])
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            latents = denoise_step(latents, ts[(args.warmup + i) % len(ts)])
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    assert torch.isfinite(latents).all(), "non-finite latents"

    el = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    if rank == 0:
        roof = measure_attention_roofline(device, dtype) if dtype == torch.bfloat16 else None
        roof_conv = measure_conv_roofline(device, dtype) if dtype == torch.bfloat16 else None
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline()
            except Exception as e:  # the baseline must never take the GPU number down with it
                cpu = {"error": repr(e)}
        f_step = unet_flops(2, HEIGHT // 8, WIDTH // 8)
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": "denoising steps/sec, 16x320x512 bf16 U-Net+CMC+OMC",
            "value": round(world * args.steps / elapsed, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "16x320x512 clip, CFG batch 2, full-width 3D U-Net (1.39B params, random init) + "
                                   "Camera Adapter (CMC) + Object Motion Control (OMC) features, DDIM step; "
                                   "configs/obj.yaml shapes; 1 clip per GPU",
                       "frames": FRAMES, "height": HEIGHT, "width": WIDTH, "guidance_scale": args.guidance,
                       "hip_graph": not args.no_graph, "parallelism": f"dp{world} (independent clips, no collective)"},
            "unet_tflop_per_step_reference_graph": round(f_step / 1e12, 3),
            "effective_tflops_per_gpu": round(f_step / 1e12 / (ms * 1e-3), 1),
            "conditioning_once_per_clip_ms": round(cond_ms, 2),
            "roofline": roof, "roofline_conv": roof_conv, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
        if args.autotune_log:
            with open(args.autotune_log, "w") as f:
                for key, use, times, calls in sorted(K.autotune_report(), key=lambda r: -r[3] * min(r[2].values() or [0])):
                    f.write(f"{key} -> arm {use}  calls {calls}  ms {times}\n")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
