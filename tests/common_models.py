"""Builders shared by the CPU and GPU tests: a reduced FMC stack (same topology as configs/obj.yaml, narrower)
instantiated twice -- the CPU oracle and the gfx950 product -- with identical seeded weights."""
import copy

import numpy as np
import torch

from oracle import conditioning as OC
from oracle import fmc_modules as OM

from synfmc_amd.configs import (MMK, adapter_kwargs, encoder_kwargs, processor_kwargs, synthetic_clip, unet_kwargs)  # noqa: E402,F401


def reseed(module, seed, std=0.05, fan_in_gain=None):
    """Seeded N(0, std) for every parameter: zero-initialised layers (qkv_merge, zero convs, LoRA up) would make the
    conditioning paths invisible otherwise.  Norm gains are centred on 1.  With `fan_in_gain` matrices / filters get
    std = gain / sqrt(fan_in): a contractive net whose GRADIENTS are well conditioned (the fixed-std net is chaotic:
    its directional derivatives differ by orders of magnitude between fp-equivalent implementations)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            s = std
            if fan_in_gain is not None and p.ndim >= 2:
                s = fan_in_gain / (p[0].numel() ** 0.5)
            v = torch.randn(p.shape, generator=g) * s
            if p.ndim == 1 and ("norm" in name and name.endswith("weight")):
                v = v + 1.0
            p.copy_(v)
    return module


def build_oracle(widths=(64, 128, 256, 256), cross_dim=64, conditioned=True, lora=True, seed=0, fan_in_gain=None,
                 enc_max_len=16, motion_lora=False):
    unet = OM.UNet3DConditionModelCamObjCond(**unet_kwargs(widths, cross_dim))
    if conditioned:
        unet.set_all_attn_processor(**processor_kwargs(widths, lora, motion_lora=motion_lora))
        OM.patch_down_blocks_for_omc(unet)
    reseed(unet, seed, fan_in_gain=fan_in_gain)
    enc = reseed(OM.CameraPoseEncoder(**encoder_kwargs(widths, enc_max_len)), seed + 1, fan_in_gain=fan_in_gain) \
        if conditioned else None
    ada = reseed(OM.Adapter(**adapter_kwargs(widths)), seed + 2, fan_in_gain=fan_in_gain) if conditioned else None
    return unet.eval(), (enc.eval() if enc else None), (ada.eval() if ada else None)


def build_lora_only(widths=(64, 128, 256, 256), cross_dim=64, seed=0, fan_in_gain=None, device=None,
                    dtype=torch.float32):
    """BASELINE configs[1]: the 3-D U-Net with the Domain LoRA on every spatial attention (attn1 and attn2), plain
    temporal attention, no camera / object conditioning.  Returns (oracle, product or None)."""
    ou = OM.UNet3DConditionModelCamObjCond(**unet_kwargs(widths, cross_dim))
    ou.set_all_attn_processor(**processor_kwargs(widths, True, temporal=False))
    reseed(ou, seed, fan_in_gain=fan_in_gain).eval()
    if device is None:
        return ou, None
    from synfmc_amd.models.unet import UNet3DConditionModel
    pu = UNet3DConditionModel(**unet_kwargs(widths, cross_dim))       # the base model, as the plain pipeline uses it
    pu.set_image_layer_lora(2)                                        # unet.py:407-421: rank = C // 2 (configs/lora.yaml)
    pu.load_state_dict(ou.state_dict(), strict=True)
    return ou, pu.to(device=device, dtype=dtype).eval().requires_grad_(False)


def build_product(oracle_unet, oracle_enc, oracle_ada, widths=(64, 128, 256, 256), cross_dim=64, conditioned=True,
                  lora=True, device="cuda", dtype=torch.float32, enc_max_len=16, motion_lora=False):
    from synfmc_amd.adapter import Adapter
    from synfmc_amd.models.pose_adaptor import CameraPoseEncoder
    from synfmc_amd.models.unet import UNet3DConditionModelCamObjCond
    from synfmc_amd.modified_modules import patch_unet_for_omc
    unet = UNet3DConditionModelCamObjCond(**unet_kwargs(widths, cross_dim))
    if conditioned:
        unet.set_all_attn_processor(**processor_kwargs(widths, lora, motion_lora=motion_lora))
        patch_unet_for_omc(unet)
    missing, unexpected = unet.load_state_dict(oracle_unet.state_dict(), strict=True)
    unet = unet.to(device=device, dtype=dtype).eval().requires_grad_(False)
    enc = ada = None
    if conditioned:
        enc = CameraPoseEncoder(**encoder_kwargs(widths, enc_max_len))
        enc.load_state_dict(oracle_enc.state_dict(), strict=True)
        enc = enc.to(device=device, dtype=dtype).eval().requires_grad_(False)
        ada = Adapter(**adapter_kwargs(widths))
        ada.load_state_dict(oracle_ada.state_dict(), strict=True)
        ada = ada.to(device=device, dtype=dtype).eval().requires_grad_(False)
    return unet, enc, ada


class bf16_rounding:
    """Context manager: run an oracle module "as bf16 storage would" -- every leaf module's output (Linear, Conv2d,
    GroupNorm, LayerNorm, activations) is rounded to bf16 and the weights are bf16-rounded for the duration, while the
    arithmetic stays the oracle's fp32.  The distance between this run and the plain fp32 oracle is the error the
    bf16 FORMAT alone introduces through the depth of the network; a bf16 kernel path can be judged against it."""

    def __init__(self, *modules):
        self.modules = [m for m in modules if m is not None]
        self.handles, self.saved = [], []

    @staticmethod
    def _round(x):
        if torch.is_tensor(x) and x.is_floating_point():
            return x.to(torch.bfloat16).to(x.dtype)
        if isinstance(x, (tuple, list)):
            return type(x)(bf16_rounding._round(v) for v in x)
        return x

    def __enter__(self):
        for m in self.modules:
            for sub in m.modules():
                if not list(sub.children()):
                    self.handles.append(sub.register_forward_hook(lambda _m, _i, out: bf16_rounding._round(out)))
            for p in m.parameters():
                self.saved.append((p, p.detach().clone()))      # the bf16 product stores every parameter in bf16
                with torch.no_grad():
                    p.copy_(p.to(torch.bfloat16).float())
        return self

    def __exit__(self, *exc):
        for h in self.handles:
            h.remove()
        with torch.no_grad():
            for p, v in self.saved:
                p.copy_(v)
        self.handles, self.saved = [], []
        return False


FULL_WIDTHS = (320, 640, 1280, 1280)
FULL_CROSS_DIM = 768


def full_width_case(seed=43, clip_seed=143, H=128, W=192, Fr=16):
    """The benchmarked architecture (SD-1.5 widths 320/640/1280/1280, head dims 40/80/160, text width 768, CMC + OMC,
    LoRA on the spatial attention) on a clip the CPU oracle finishes in seconds.  Weights are fan-in scaled
    (`reseed(..., fan_in_gain=1.0)`): a variance-preserving net, so rel-inf errors measure arithmetic and not chaos."""
    ou, oe, oa = build_oracle(FULL_WIDTHS, FULL_CROSS_DIM, seed=seed, fan_in_gain=1.0)
    clip = synthetic_clip(B=1, Fr=Fr, H=H, W=W, cross_dim=FULL_CROSS_DIM, seed=clip_seed)
    return ou, oe, oa, clip
