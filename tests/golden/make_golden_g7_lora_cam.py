#!/usr/bin/env python
"""G7: ONE denoising step of BASELINE.json configs[1] (Domain LoRA only, configs/lora.yaml) and configs[2] (CMC: Camera Encoder + Adapter,
configs/cam.yaml) at FULL width on the full 16x320x512 clip at classifier-free-guidance batch 2 -- the reference's OWN code:
  * configs[1]: `fmc.models.unet.UNet3DConditionModel` + `set_image_layer_lora(2)` (`/root/reference/fmc/models/unet.py:288-311`: LoRA of rank
    C / 2 on every spatial attn1 / attn2), fed as `AnimationPipeline.__call__` feeds it (`cat([latents] * 2)`, `[uncond, text]`);
  * configs[2]: `fmc.models.unet.UNet3DConditionModelPoseCond` + `fmc.models.pose_adaptor.CameraPoseEncoder`, fed as `CameraCtrlPipeline` does
    (pose features duplicated for the two halves);
over the restated diffusers primitives (see make_golden_g5.py for the shim).  Weights / inputs come from seeds (`tests/common_models`): only
seeds and outputs are stored, the GPU test re-generates the weights.  The oracle is run on the same inputs and must agree (printed and stored).
Build container only (the reference is not on the GPU box); writes data only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_g7_lora_cam.py
"""
import os
import sys
import time

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from tests.golden.make_golden import install_stubs                 # noqa: E402
from tests.golden.make_golden_g5 import install_diffusers_shim     # noqa: E402

LORA_SEED, CAM_SEED, CLIP_SEED, H, W, T_STEP, UNCOND_SEED = 53, 43, 143, 320, 512, 801, 99


def install_diffusers_lora_processor():
    """`set_image_layer_lora` (fmc/models/unet.py:288-311) instantiates DIFFUSERS' own `LoRAAttnProcessor`, which the G5 shim leaves out (the
    stage-2/3 models use fmc's `LoRAAttnProcessor`).  diffusers==0.24.0 (environment.yaml:13), `models/attention_processor.py`: the class holds
    four `LoRALinearLayer`s and, on its first call, hangs them on the attention's projections as `lora_layer`, swaps itself for the stock
    `AttnProcessor` and calls it; `LoRACompatibleLinear.forward(x, scale)` = `linear(x) + scale * lora_layer(x)` (`models/lora.py`)."""
    from torch import nn
    from oracle import diffusers_restated as OD

    def lora_linear_forward(self, hidden_states, scale: float = 1.0):
        y = nn.Linear.forward(self, hidden_states)
        layer = getattr(self, "lora_layer", None)
        return y if layer is None else y + scale * layer(hidden_states)
    OD.LoRACompatibleLinear.forward = lora_linear_forward           # (this process only: the restatement itself never carries a lora_layer)

    class LoRAAttnProcessor(nn.Module):
        def __init__(self, hidden_size, cross_attention_dim=None, rank=4, network_alpha=None, **kwargs):
            super().__init__()
            self.hidden_size, self.cross_attention_dim, self.rank = hidden_size, cross_attention_dim, rank
            self.to_q_lora = OD.LoRALinearLayer(hidden_size, hidden_size, rank, network_alpha)
            self.to_k_lora = OD.LoRALinearLayer(cross_attention_dim or hidden_size, hidden_size, rank, network_alpha)
            self.to_v_lora = OD.LoRALinearLayer(cross_attention_dim or hidden_size, hidden_size, rank, network_alpha)
            self.to_out_lora = OD.LoRALinearLayer(hidden_size, hidden_size, rank, network_alpha)

        def __call__(self, attn, hidden_states, *args, **kwargs):
            attn.to_q.lora_layer = self.to_q_lora.to(hidden_states.device)
            attn.to_k.lora_layer = self.to_k_lora.to(hidden_states.device)
            attn.to_v.lora_layer = self.to_v_lora.to(hidden_states.device)
            attn.to_out[0].lora_layer = self.to_out_lora.to(hidden_states.device)
            attn._modules.pop("processor")
            attn.processor = OD.DefaultAttnProcessor()
            return attn.processor(attn, hidden_states, *args, **kwargs)
    sys.modules["diffusers.models.attention_processor"].LoRAAttnProcessor = LoRAAttnProcessor


def main():
    install_stubs()
    install_diffusers_shim()
    install_diffusers_lora_processor()
    sys.path.insert(0, REF)
    from einops import rearrange
    from tests import common_models as CM
    from oracle import conditioning as OC
    from fmc.models.unet import UNet3DConditionModel, UNet3DConditionModelPoseCond
    from fmc.models.pose_adaptor import CameraPoseEncoder
    WF, XD = CM.FULL_WIDTHS, CM.FULL_CROSS_DIM
    clip = CM.synthetic_clip(B=1, Fr=16, H=H, W=W, cross_dim=XD, seed=CLIP_SEED)
    g = torch.Generator().manual_seed(UNCOND_SEED)
    text2 = torch.cat([torch.randn(1, 77, XD, generator=g), clip["text"]])
    x2 = torch.cat([clip["latents"], clip["latents"]])
    t = torch.tensor(T_STEP)
    out = {}
    with torch.no_grad():
        # ---- configs[1]: Domain LoRA only ------------------------------------------------------------------------------------------
        ou, _ = CM.build_lora_only(WF, XD, seed=LORA_SEED, fan_in_gain=1.0)
        rb = UNet3DConditionModel(**CM.unet_kwargs(WF, XD))
        rb.set_image_layer_lora(2)
        assert sorted(rb.state_dict().keys()) == sorted(ou.state_dict().keys())
        rb.load_state_dict(ou.state_dict(), strict=True)
        rb.eval()
        t0 = time.time()
        eps_ref = rb(x2, t, text2).sample
        print(f"configs[1] reference code, CFG-2 16x{H}x{W}: {time.time() - t0:.1f} s", flush=True)
        del rb
        t0 = time.time()
        eps_ora = ou(x2, t, text2).sample
        print(f"configs[1] oracle: {time.time() - t0:.1f} s", flush=True)
        err = float((eps_ref - eps_ora).abs().max() / eps_ref.abs().max())
        print("configs[1] reference vs oracle rel-inf:", err, flush=True)
        assert err < 1e-5
        out.update(lora_eps=eps_ref.numpy().astype(np.float32), lora_oracle_vs_reference=np.array(err))
        del ou, eps_ref, eps_ora
        # ---- configs[2]: CMC (camera encoder + adapter), no OMC --------------------------------------------------------------------
        ou, oe, oa, clip2 = CM.full_width_case(CAM_SEED, CLIP_SEED, H, W)
        assert torch.equal(clip2["latents"], clip["latents"])
        rp = UNet3DConditionModelPoseCond(**CM.unet_kwargs(WF, XD))
        rp.set_all_attn_processor(**CM.processor_kwargs(WF, True))
        rp.load_state_dict(ou.state_dict(), strict=True)
        re_ = CameraPoseEncoder(**CM.encoder_kwargs(WF))
        re_.load_state_dict(oe.state_dict(), strict=True)
        rp.eval(); re_.eval()
        pose_emb = rearrange(OC.to_plucker_embedding(clip["c2w"], clip["K"], (H, W)), "b f c h w -> b c f h w")
        pose2 = [torch.cat([x, x]) for x in (rearrange(y, "(b f) c h w -> b c f h w", b=1) for y in re_(pose_emb))]
        t0 = time.time()
        eps_ref = rp(x2, t, text2, pose_embedding_features=pose2).sample
        print(f"configs[2] reference code: {time.time() - t0:.1f} s", flush=True)
        del rp, re_
        pose2o = [torch.cat([x, x]) for x in (rearrange(y, "(b f) c h w -> b c f h w", b=1) for y in oe(pose_emb))]
        t0 = time.time()
        eps_ora = ou(x2, t, text2, pose_embedding_features=pose2o, traj_features=None).sample
        print(f"configs[2] oracle: {time.time() - t0:.1f} s", flush=True)
        err = float((eps_ref - eps_ora).abs().max() / eps_ref.abs().max())
        print("configs[2] reference vs oracle rel-inf:", err, flush=True)
        assert err < 1e-5
        out.update(cam_eps=eps_ref.numpy().astype(np.float32), cam_oracle_vs_reference=np.array(err))
    np.savez_compressed(os.path.join(HERE, "g7_lora_cam_steps.npz"), lora_seed=np.array(LORA_SEED), cam_seed=np.array(CAM_SEED),
                        clip_seed=np.array(CLIP_SEED), uncond_seed=np.array(UNCOND_SEED), hw=np.array([H, W]), t=np.array(T_STEP), **out)
    print("G7 written", flush=True)


if __name__ == "__main__":
    main()
