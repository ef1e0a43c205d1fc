#!/usr/bin/env python
"""G6: ONE denoising step of the BENCHMARKED configuration (BASELINE.json configs[3], the metric's workload): the reference's
OWN `UNet3DConditionModelCamObjCond` + `CameraPoseEncoder` (`/root/reference/fmc/models/unet_cam_obj.py:1107-1375`, forward
monkey-patch as `train_cam_obj_ctrl.py:317-329`) at SD-1.5 widths, on the FULL 16x320x512 clip at classifier-free-guidance
batch 2 exactly as `CameraObjCtrlPipeline.__call__` feeds it (`fmc/pipelines/pipeline_animation_cm_om.py:668-676,704`:
latents and pose features duplicated, OMC features zero for the unconditional half), over the restated diffusers primitives
(see make_golden_g5.py for the shim).  Weights / inputs: `tests/common_models.full_width_case(seed, clip_seed, 320, 512)` --
only seeds and outputs are stored, the GPU test re-generates the 1.7 B weights from the seed.  The oracle is run on the same
inputs and must agree (printed; tests/test_oracle_golden.py cannot repeat a 16x320x512 CPU forward inside the CPU suite's
budget, so THIS script is where the oracle is pinned at full size).  Build container only; writes data only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_g6_bench_step.py
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

SEED, CLIP_SEED, H, W, T_STEP, UNCOND_SEED = 43, 143, 320, 512, 801, 99


def cfg_inputs(clip, pose_feats, traj):
    """What the pipeline hands the U-Net for one CFG step (pipeline_animation_cm_om.py:668-676,704)."""
    g = torch.Generator().manual_seed(UNCOND_SEED)
    uncond = torch.randn(1, 77, clip["text"].shape[-1], generator=g)
    text2 = torch.cat([uncond, clip["text"]])
    x2 = torch.cat([clip["latents"], clip["latents"]])
    pose2 = [torch.cat([x, x]) for x in pose_feats]
    traj2 = [torch.cat([torch.zeros_like(x), x]) for x in traj]
    return x2, text2, pose2, traj2


def main():
    install_stubs()
    install_diffusers_shim()
    sys.path.insert(0, REF)
    from einops import rearrange
    from tests import common_models as CM
    from oracle import conditioning as OC
    from fmc.models.unet_cam_obj import UNet3DConditionModelCamObjCond
    from fmc.models.pose_adaptor import CameraPoseEncoder
    from fmc.modified_modules import Adapted_CrossAttnDownBlock3D_forward, Adapted_DownBlock3D_forward

    ou, oe, oa, clip = CM.full_width_case(SEED, CLIP_SEED, H, W)
    WF, XD = CM.FULL_WIDTHS, CM.FULL_CROSS_DIM
    ru = UNet3DConditionModelCamObjCond(**CM.unet_kwargs(WF, XD))
    ru.set_all_attn_processor(**CM.processor_kwargs(WF, True))
    idx = 0
    for name, module in ru.down_blocks.named_modules():
        cls = module.__class__.__name__
        if cls in ("CrossAttnDownBlock3D", "DownBlock3D"):
            fwd = Adapted_CrossAttnDownBlock3D_forward if cls == "CrossAttnDownBlock3D" else Adapted_DownBlock3D_forward
            setattr(module, "forward", fwd.__get__(module, module.__class__))
            setattr(module, "traj_fea_idx", idx)
            idx += 1
    ru.load_state_dict(ou.state_dict(), strict=True)
    re_ = CameraPoseEncoder(**CM.encoder_kwargs(WF))
    re_.load_state_dict(oe.state_dict(), strict=True)
    ru.eval(); re_.eval()
    with torch.no_grad():
        pose_emb = rearrange(OC.to_plucker_embedding(clip["c2w"], clip["K"], (H, W)), "b f c h w -> b c f h w")
        pf_ref = re_(pose_emb)
        pose_feats = [rearrange(x, "(b f) c h w -> b c f h w", b=1) for x in pf_ref]
        traj = OC.get_traj_features(clip["infos"], clip["masks"], oa)
        x2, text2, pose2, traj2 = cfg_inputs(clip, pose_feats, traj)
        t = torch.tensor(T_STEP)
        t0 = time.time()
        eps_ref = ru(x2, t, text2, pose_embedding_features=pose2, traj_features=traj2).sample
        print(f"reference code, CFG-2 16x{H}x{W}: {time.time() - t0:.1f} s")
        del ru
        t0 = time.time()
        pose_o = [rearrange(x, "(b f) c h w -> b c f h w", b=1) for x in oe(pose_emb)]
        x2o, text2o, pose2o, traj2o = cfg_inputs(clip, pose_o, traj)
        eps_ora = ou(x2o, t, text2o, pose_embedding_features=pose2o, traj_features=traj2o).sample
        print(f"oracle: {time.time() - t0:.1f} s")
        # the same step with every layer output and weight of the oracle rounded to bf16 (common_models.bf16_rounding): what the FORMAT alone
        # costs on this case.  The GPU test asserts that the bf16 kernel path stays within 2x of it from this tensor (round 4: at full size the
        # only bf16 assert was a max-norm bound; the assert that isolates kernel error from format error existed at 16x128x192 only).
        t0 = time.time()
        with CM.bf16_rounding(ou, oe, oa):
            pose16 = [rearrange(x, "(b f) c h w -> b c f h w", b=1) for x in oe(pose_emb)]
            traj16 = OC.get_traj_features(clip["infos"], clip["masks"], oa)
            x2h, text2h, pose2h, traj2h = cfg_inputs(clip, pose16, traj16)
            eps16 = ou(x2h, t, text2h, pose_embedding_features=pose2h, traj_features=traj2h).sample
        print(f"bf16-rounded oracle: {time.time() - t0:.1f} s")
    err = float((eps_ref - eps_ora).abs().max() / eps_ref.abs().max())
    print("reference vs oracle rel-inf:", err)
    assert err < 1e-5
    fmt = float((eps16 - eps_ref).abs().max() / eps_ref.abs().max())
    print("bf16 format error (rounded oracle vs reference) rel-inf:", fmt)
    np.savez_compressed(os.path.join(HERE, "g6_bench_step.npz"), seed=np.array(SEED), clip_seed=np.array(CLIP_SEED),
                        uncond_seed=np.array(UNCOND_SEED), hw=np.array([H, W]), t=np.array(T_STEP),
                        eps=eps_ref.numpy().astype(np.float32), oracle_vs_reference=np.array(err),
                        eps_bf16_rounded_oracle=eps16.numpy().astype(np.float32), bf16_format_err=np.array(fmt),
                        enc_feat_sums=np.array([float(x.double().sum()) for x in pf_ref]))
    print("G6 written:", tuple(eps_ref.shape), float(eps_ref.abs().max()))


if __name__ == "__main__":
    main()
