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
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None


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
    other = type("_OtherScheduler", (), {})
    _mod("diffusers.schedulers", DDIMScheduler=OD.DDIMScheduler, PNDMScheduler=other, LMSDiscreteScheduler=other,
         EulerDiscreteScheduler=other, EulerAncestralDiscreteScheduler=other, DPMSolverMultistepScheduler=other)
    sys.modules["diffusers.models"].AutoencoderKL = type("AutoencoderKL", (), {})

    class DiffusionPipeline:
        _optional_components = []

        def register_modules(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

        def progress_bar(self, iterable=None, total=None):
            import contextlib

            class _Bar:
                def update(self, n=1):
                    pass

            @contextlib.contextmanager
            def cm():
                yield _Bar()
            return cm() if iterable is None else iterable

    _mod("transformers", CLIPTextModel=type("CLIPTextModel", (), {}), CLIPTokenizer=type("CLIPTokenizer", (), {}))   # type hints only
    _mod("diffusers.pipelines")
    _mod("diffusers.pipelines.pipeline_utils", DiffusionPipeline=DiffusionPipeline)


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
    # ---- the reference's own denoising loop (CameraObjCtrlPipeline.__call__, pipeline_animation_cm_om.py:570-738):
    # CFG concatenation order, zero traj features for the unconditional half, omcm_min_step gating, scheduler.step.
    # VAE / CLIP are outside the path: stub tokenizer + text encoder map a prompt to fixed embeddings, the stub VAE decodes
    # linearly so the final latents can be read back from the returned "video".
    from fmc.pipelines.pipeline_animation_cm_om import CameraObjCtrlPipeline
    emb_text, emb_uncond = clip["text"], torch.randn(1, 77, 64, generator=torch.Generator().manual_seed(7))

    class _Tok:
        model_max_length = 77

        def __call__(self, prompt, **kw):
            ids = torch.tensor([[1 if p else 0] * 77 for p in (prompt if isinstance(prompt, list) else [prompt])])
            return types.SimpleNamespace(input_ids=ids, attention_mask=None)

        def batch_decode(self, ids):
            return [""]

    class _TextEnc(nn.Module):
        config = types.SimpleNamespace()
        dtype = torch.float32

        def forward(self, ids, attention_mask=None):
            return (torch.cat([emb_text if int(r[0]) else emb_uncond for r in ids], 0),)

    class _Vae:
        config = types.SimpleNamespace(block_out_channels=[1, 1, 1, 1])

        def decode(self, z):
            return types.SimpleNamespace(sample=z * 0.18215 * 0.01)   # keeps |latent| < 100 inside the clamp

    sched = OD.DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear",
                             steps_offset=1, clip_sample=False)
    sched.config = types.SimpleNamespace(steps_offset=1, clip_sample=False)
    object.__setattr__(ru, "in_channels", 4)
    pipe = CameraObjCtrlPipeline(_Vae(), _TextEnc(), _Tok(), ru, sched, re_)
    pipe_out = {}
    for name, gs, min_step in (("cfg2_gate700", 2.0, 700), ("cfg2_nogate", 2.0, 0), ("nocfg", 1.0, 0)):
        video = pipe("a prompt", pose_emb, video_length=16, traj_features=[x.clone() for x in traj], height=128, width=128,
                     num_inference_steps=6, guidance_scale=gs, negative_prompt=None, latents=clip["latents"].clone(),
                     output_type="tensor", omcm_min_step=min_step, multidiff_overlaps=0).videos
        pipe_out[name] = ((video.double() - 0.5) * 200.0).float().numpy()   # undo the stub VAE: final latents
    np.savez_compressed(os.path.join(HERE, "g5_pipeline.npz"), emb_uncond=emb_uncond.numpy(), **pipe_out)
    np.savez_compressed(os.path.join(HERE, "g5_unet_variants.npz"), out_posecond=out_posecond.numpy(),
                        out_base=out_base.numpy(), base_seed=np.array(41))
    np.savez_compressed(os.path.join(HERE, "g5_unet_cmc_omc.npz"),
                        widths=np.array(W4), seed=np.array(40), clip_seed=np.array(140),
                        out=out.numpy(), out_notraj=out_notraj.numpy(), out_t17=out_t2.numpy(),
                        enc_feat_sums=np.array([float(x.double().sum()) for x in pf_ref]),
                        enc_feat0=pf_ref[0][:2, :8].numpy(), enc_feat3=pf_ref[3][:2, :8].numpy(),
                        n_keys=np.array(len(ref_keys)))
    # ---- the same reference forward at the widths of the GPU parity tests (head dims the HIP kernels accept), so that
    # tests/test_gpu_model.py can compare the product with the reference's output directly
    WG = (64, 128, 256, 256)
    og, oeg, oag = CM.build_oracle(WG, seed=42)
    rg = UNet3DConditionModelCamObjCond(**CM.unet_kwargs(WG, 64))
    rg.set_all_attn_processor(**CM.processor_kwargs(WG, True))
    idx = 0
    for name, module in rg.down_blocks.named_modules():
        cls = module.__class__.__name__
        if cls in ("CrossAttnDownBlock3D", "DownBlock3D"):
            fwd = Adapted_CrossAttnDownBlock3D_forward if cls == "CrossAttnDownBlock3D" else Adapted_DownBlock3D_forward
            setattr(module, "forward", fwd.__get__(module, module.__class__))
            setattr(module, "traj_fea_idx", idx)
            idx += 1
    rg.load_state_dict(og.state_dict(), strict=True)
    reg = CameraPoseEncoder(**CM.encoder_kwargs(WG))
    reg.load_state_dict(oeg.state_dict(), strict=True)
    rg.eval(); reg.eval()
    with torch.no_grad():
        pfg = [rearrange(x, "(b f) c h w -> b c f h w", b=1) for x in reg(pose_emb)]
        trajg = OC.get_traj_features(clip["infos"], clip["masks"], oag)
        outg = rg(clip["latents"], t, clip["text"], pose_embedding_features=pfg, traj_features=trajg).sample
    np.savez_compressed(os.path.join(HERE, "g5_unet_gpu_widths.npz"), widths=np.array(WG), seed=np.array(42),
                        clip_seed=np.array(140), out=outg.numpy())
    with open(os.path.join(HERE, "g5_unet_keys.txt"), "w") as f:
        f.write("\n".join(ref_keys) + "\n")
    print("G5 written:", out.shape, float(out.abs().max()), "keys", len(ref_keys))


if __name__ == "__main__":
    main()
