#!/usr/bin/env python
"""G5 at the benchmarked widths: the reference's OWN `UNet3DConditionModelCamObjCond` + `CameraPoseEncoder`
(`/root/reference/fmc/models/unet_cam_obj.py:1107-1375`, forward monkey-patch as `train_cam_obj_ctrl.py:317-329`)
at SD-1.5 widths 320/640/1280/1280 (head dims 40/80/160), text width 768, CMC + OMC, over the restated diffusers
primitives (see make_golden_g5.py for the shim and for what this does and does not pin).  Weights and inputs are
`tests/common_models.full_width_case()` (seeded; only seeds + outputs are stored -- 1.7 B weights are re-generated
from the seed by the tests).  Build container only (imports /root/reference); writes data only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_g5_full_width.py
"""
import os
import sys

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from tests.golden.make_golden import install_stubs                 # noqa: E402
from tests.golden.make_golden_g5 import install_diffusers_shim     # noqa: E402


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

    seed, clip_seed, H, W = 43, 143, 128, 192
    ou, oe, oa, clip = CM.full_width_case(seed, clip_seed, H, W)
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
    n_params = sum(p.numel() for p in ru.parameters())
    with torch.no_grad():
        plucker = OC.to_plucker_embedding(clip["c2w"], clip["K"], (H, W))
        pose_emb = rearrange(plucker, "b f c h w -> b c f h w")
        pf_ref = re_(pose_emb)
        pose_feats = [rearrange(x, "(b f) c h w -> b c f h w", b=1) for x in pf_ref]
        traj = OC.get_traj_features(clip["infos"], clip["masks"], oa)
        t = torch.tensor([801])
        out = ru(clip["latents"], t, clip["text"], pose_embedding_features=pose_feats, traj_features=traj).sample
        out_notraj = ru(clip["latents"], t, clip["text"], pose_embedding_features=pose_feats, traj_features=None).sample
        # the oracle on the same inputs, for the record (tests/test_oracle_golden.py re-checks it)
        ora = ou(clip["latents"], t, clip["text"],
                 pose_embedding_features=[rearrange(x, "(b f) c h w -> b c f h w", b=1) for x in oe(pose_emb)],
                 traj_features=traj).sample
    print("reference vs oracle rel-inf:", float((out - ora).abs().max() / out.abs().max()))
    np.savez_compressed(os.path.join(HERE, "g5_unet_full_width.npz"), widths=np.array(WF), cross_dim=np.array(XD),
                        seed=np.array(seed), clip_seed=np.array(clip_seed), hw=np.array([H, W]),
                        out=out.numpy(), out_notraj=out_notraj.numpy(), n_params=np.array(n_params),
                        enc_feat_sums=np.array([float(x.double().sum()) for x in pf_ref]))
    print("G5 full width written:", out.shape, float(out.abs().max()), "params", n_params)


if __name__ == "__main__":
    main()
