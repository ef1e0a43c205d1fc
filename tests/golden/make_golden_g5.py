#!/usr/bin/env python
"""G5: the reference's OWN U-Net-level modules executed here, over restated diffusers primitives.

`fmc/models/*.py` and `fmc/modified_modules.py` hard-import `diffusers==0.24.0`, which is not installable in the build
container.  This script registers in-memory stand-in modules named `diffusers.*` whose symbols are the primitives of
`oracle/diffusers_restated.py` (ResnetBlock2D, Transformer2DModel, Attention, FeedForward, LoRALinearLayer, Timesteps,
...) plus inert infrastructure (`ModelMixin`, `ConfigMixin`, `register_to_config`, `BaseOutput`, logging), then imports
the reference and runs ITS code:

  * `fmc.models.unet_cam_obj.UNet3DConditionModelCamObjCond` (forward order, kwargs routing, skip arithmetic),
    `set_all_attn_processor` (processor registry), `fmc.models.attention_processor.*` (Camera-Adapter merge, LoRA),
    `fmc.models.motion_module.*` (PE placement, temporal blocks), `fmc.models.unet_blocks.*`,
    `fmc.modified_modules.Adapted_*_forward` patched exactly as `train_cam_obj_ctrl.py:317-329` does,
    `fmc.models.pose_adaptor.CameraPoseEncoder`, `fmc.adapter.Adapter`;
  * weights = the oracle's seeded state dicts, loaded with strict=True  ->  pins the state-dict key set as well;
  * inputs = `tests/common_models.synthetic_clip` (seeded), outputs stored in `tests/golden/g5_*.npz`.

What this pins: every line of the reference-owned `fmc/` modules on the path.  What it does NOT pin: the diffusers
primitives themselves (still the restatement).  `tests/test_oracle_golden.py` then checks the
oracle (`oracle/fmc_modules.py`) against these vectors.  Only data is written; run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_g5.py
"""
import dataclasses
import logging as pylogging
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch
from torch import nn

from oracle import diffusers_restated as OD          # noqa: E402
from tests.golden.make_golden import install_stubs    # noqa: E402  (decord / cv2 / nltk / imageio / torchvision)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent:
        setattr(sys.modules[parent], leaf, m)
    return m


class _Config(dict):
    __getattr__ = dict.__getitem__


def register_to_config(init):
    import functools
    import inspect

    @functools.wraps(init)
    def wrapper(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        object.__setattr__(self, "_cfg", _Config(cfg))
        init(self, *args, **kwargs)
    return wrapper


class ConfigMixin:
    config_name = "config.json"

    @property
    def config(self):
        return self._cfg


class ModelMixin(nn.Module):
    _supports_gradient_checkpointing = False

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


class BaseOutput:
    """dataclass-style output container (`.sample`)"""
    def __getitem__(self, i):
        return dataclasses.astuple(self)[i]


class AttnProcsLayers(nn.Module):
    def __init__(self, procs):
        super().__init__()
        self.layers = nn.ModuleList([p for p in procs.values() if isinstance(p, nn.Module)])


class _Unused(nn.Module):                         # AdaGroupNorm / SpatialNorm / diffusers' own LoRAAttnProcessor: names only
    def __init__(self, *a, **k):
        raise NotImplementedError("not on the FMC path")


def install_diffusers_shim():
    lg = types.SimpleNamespace(get_logger=lambda name=None: pylogging.getLogger(name or "fmc"))
    _mod("diffusers")
    _mod("diffusers.utils", BaseOutput=BaseOutput, logging=lg, USE_PEFT_BACKEND=False,
         deprecate=lambda *a, **k: None, is_accelerate_available=lambda: False,
         SAFETENSORS_WEIGHTS_NAME="diffusion_pytorch_model.safetensors", WEIGHTS_NAME="diffusion_pytorch_model.bin")
    _mod("diffusers.utils.import_utils", is_xformers_available=lambda: False)
    _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config, FrozenDict=_Config)
    _mod("diffusers.models", ModelMixin=ModelMixin)
    _mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    _mod("diffusers.models.lora", LoRALinearLayer=OD.LoRALinearLayer, LoRACompatibleLinear=OD.LoRACompatibleLinear,
         LoRACompatibleConv=OD.LoRACompatibleConv)
    _mod("diffusers.models.attention_processor", Attention=OD.Attention, AttentionProcessor=object,
         LoRAAttnProcessor=_Unused, SpatialNorm=_Unused, AttnProcessor=OD.DefaultAttnProcessor)
    _mod("diffusers.models.attention", Attention=OD.Attention, FeedForward=OD.FeedForward,
         BasicTransformerBlock=OD.BasicTransformerBlock)
    _mod("diffusers.models.resnet", ResnetBlock2D=OD.ResnetBlock2D, Downsample2D=OD.Downsample2D, Upsample2D=OD.Upsample2D)
    _mod("diffusers.models.transformer_2d", Transformer2DModel=OD.Transformer2DModel)
    _mod("diffusers.models.embeddings", TimestepEmbedding=OD.TimestepEmbedding, Timesteps=OD.Timesteps)
    _mod("diffusers.models.activations", get_activation=OD.get_activation)
    _mod("diffusers.models.normalization", AdaGroupNorm=_Unused)
    _mod("diffusers.loaders", AttnProcsLayers=AttnProcsLayers, UNet2DConditionLoadersMixin=type("UNet2DConditionLoadersMixin", (), {}),
         LoraLoaderMixin=type("LoraLoaderMixin", (), {}))
    _mod("diffusers.schedulers", DDIMScheduler=OD.DDIMScheduler)


def main():
    install_stubs()
    install_diffusers_shim()
    sys.path.insert(0, REF)
    from einops import rearrange
    from tests import common_models as CM
    from oracle import conditioning as OC
    from fmc.models.unet_cam_obj import UNet3DConditionModelCamObjCond
    from fmc.models.pose_adaptor import CameraPoseEncoder
    from fmc.adapter import Adapter
    from fmc.modified_modules import Adapted_CrossAttnDownBlock3D_forward, Adapted_DownBlock3D_forward

    W4 = (32, 64, 64, 64)
    torch.manual_seed(0)
    ou, oe, oa = CM.build_oracle(W4, seed=40)                       # the oracle supplies the (seeded) weights
    ru = UNet3DConditionModelCamObjCond(**CM.unet_kwargs(W4, 64))
    ru.set_all_attn_processor(**CM.processor_kwargs(W4, True))
    # the forward monkey-patch, as train_cam_obj_ctrl.py:317-329
    idx = 0
    for name, module in ru.down_blocks.named_modules():
        cls = module.__class__.__name__
        if cls == "CrossAttnDownBlock3D":
            setattr(module, "forward", Adapted_CrossAttnDownBlock3D_forward.__get__(module, module.__class__))
            setattr(module, "traj_fea_idx", idx)
            idx += 1
        elif cls == "DownBlock3D":
            setattr(module, "forward", Adapted_DownBlock3D_forward.__get__(module, module.__class__))
            setattr(module, "traj_fea_idx", idx)
            idx += 1
    ref_keys = sorted(ru.state_dict().keys())
    ora_keys = sorted(ou.state_dict().keys())
    assert ref_keys == ora_keys, (set(ref_keys) ^ set(ora_keys))
    ru.load_state_dict(ou.state_dict(), strict=True)
    re_ = CameraPoseEncoder(**CM.encoder_kwargs(W4))
    re_.load_state_dict(oe.state_dict(), strict=True)
    ra = Adapter(**CM.adapter_kwargs(W4))
    ra.load_state_dict(oa.state_dict(), strict=True)
    ru.eval(); re_.eval(); ra.eval()

    clip = CM.synthetic_clip(B=1, Fr=16, H=128, W=128, seed=140)
    with torch.no_grad():
        plucker = OC.to_plucker_embedding(clip["c2w"], clip["K"], (128, 128))
        pose_emb = rearrange(plucker, "b f c h w -> b c f h w")
        pf_ref = re_(pose_emb)                                      # reference encoder: list of (b f) c h w
        pose_feats = [rearrange(x, "(b f) c h w -> b c f h w", b=1) for x in pf_ref]
        traj = OC.get_traj_features(clip["infos"], clip["masks"], oa)      # (Adapter itself is pinned by G2)
        t = torch.tensor([801])
        out = ru(clip["latents"], t, clip["text"], pose_embedding_features=pose_feats, traj_features=traj).sample
        out_notraj = ru(clip["latents"], t, clip["text"], pose_embedding_features=pose_feats, traj_features=None).sample
        t2 = torch.tensor([17])
        out_t2 = ru(clip["latents"], t2, clip["text"], pose_embedding_features=pose_feats, traj_features=traj).sample
    # CMC-only model of fmc/models/unet.py (CameraCtrlPipeline, configs/cam.yaml), un-patched blocks, same weights
    from fmc.models.unet import UNet3DConditionModelPoseCond, UNet3DConditionModel as BaseUNet
    rp = UNet3DConditionModelPoseCond(**CM.unet_kwargs(W4, 64))
    rp.set_all_attn_processor(**CM.processor_kwargs(W4, True))
    rp.load_state_dict(ou.state_dict(), strict=True)
    rp.eval()
    # plain AnimateDiff-style base model (no processors installed): weights of an un-conditioned oracle
    ob, _, _ = CM.build_oracle(W4, conditioned=False, seed=41)
    rb = BaseUNet(**CM.unet_kwargs(W4, 64))
    assert sorted(rb.state_dict().keys()) == sorted(ob.state_dict().keys())
    rb.load_state_dict(ob.state_dict(), strict=True)
    rb.eval()
    with torch.no_grad():
        out_posecond = rp(clip["latents"], t, clip["text"], pose_embedding_features=pose_feats).sample
        out_base = rb(clip["latents"], t, clip["text"]).sample
    np.savez_compressed(os.path.join(HERE, "g5_unet_variants.npz"), out_posecond=out_posecond.numpy(),
                        out_base=out_base.numpy(), base_seed=np.array(41))
    np.savez_compressed(os.path.join(HERE, "g5_unet_cmc_omc.npz"),
                        widths=np.array(W4), seed=np.array(40), clip_seed=np.array(140),
                        out=out.numpy(), out_notraj=out_notraj.numpy(), out_t17=out_t2.numpy(),
                        enc_feat_sums=np.array([float(x.double().sum()) for x in pf_ref]),
                        enc_feat0=pf_ref[0][:2, :8].numpy(), enc_feat3=pf_ref[3][:2, :8].numpy(),
                        n_keys=np.array(len(ref_keys)))
    with open(os.path.join(HERE, "g5_unet_keys.txt"), "w") as f:
        f.write("\n".join(ref_keys) + "\n")
    print("G5 written:", out.shape, float(out.abs().max()), "keys", len(ref_keys))


if __name__ == "__main__":
    main()
