#!/usr/bin/env python
"""G7b: what the bf16 FORMAT alone costs on the two G7 cases (BASELINE configs[1] Domain LoRA, configs[2] CMC; full width, 16x320x512, CFG batch 2):
the oracle -- which `make_golden_g7_lora_cam.py` showed to reproduce the reference's own code on these exact weights and inputs to 1e-5 -- run once more
with every layer output and every weight rounded to bf16 (`tests/common_models.bf16_rounding`).  The GPU tests then bound the bf16 kernel path by a
multiple of THIS case's own format error instead of a constant borrowed from configs[3].  No reference import is
needed (the oracle is the checker); seeds and recipe are G7's.  Writes data only:

    python tests/golden/make_golden_g7_bf16_format.py
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def main():
    from einops import rearrange
    from tests import common_models as CM
    from oracle import conditioning as OC
    g7 = np.load(os.path.join(HERE, "g7_lora_cam_steps.npz"))
    H, W = (int(v) for v in g7["hw"])
    WF, XD = CM.FULL_WIDTHS, CM.FULL_CROSS_DIM
    clip = CM.synthetic_clip(B=1, Fr=16, H=H, W=W, cross_dim=XD, seed=int(g7["clip_seed"]))
    g = torch.Generator().manual_seed(int(g7["uncond_seed"]))
    text2 = torch.cat([torch.randn(1, 77, XD, generator=g), clip["text"]])
    x2 = torch.cat([clip["latents"], clip["latents"]])
    t = torch.tensor(int(g7["t"]))
    rb = lambda v: v.to(torch.bfloat16).float()
    out = {}
    with torch.no_grad():
        ou, _ = CM.build_lora_only(WF, XD, seed=int(g7["lora_seed"]), fan_in_gain=1.0)
        t0 = time.time()
        with CM.bf16_rounding(ou):
            eps16 = ou(rb(x2), t, rb(text2)).sample
        ref = torch.from_numpy(g7["lora_eps"])
        fmt = float((eps16 - ref).abs().max() / ref.abs().max())
        print(f"configs[1] bf16-rounded oracle: {time.time() - t0:.1f} s, format error vs the reference golden {fmt:.3e}", flush=True)
        out.update(lora_eps_bf16_rounded_oracle=eps16.numpy().astype(np.float32), lora_bf16_format_err=np.array(fmt))
        del ou, eps16
        ou, oe, oa, clip2 = CM.full_width_case(int(g7["cam_seed"]), int(g7["clip_seed"]), H, W)
        assert torch.equal(clip2["latents"], clip["latents"])
        pose_emb = rearrange(OC.to_plucker_embedding(clip["c2w"], clip["K"], (H, W)), "b f c h w -> b c f h w")
        t0 = time.time()
        with CM.bf16_rounding(ou, oe):
            pose2 = [torch.cat([x, x]) for x in (rearrange(y, "(b f) c h w -> b c f h w", b=1) for y in oe(pose_emb))]
            eps16 = ou(rb(x2), t, rb(text2), pose_embedding_features=pose2, traj_features=None).sample
        ref = torch.from_numpy(g7["cam_eps"])
        fmt = float((eps16 - ref).abs().max() / ref.abs().max())
        print(f"configs[2] bf16-rounded oracle: {time.time() - t0:.1f} s, format error vs the reference golden {fmt:.3e}", flush=True)
        out.update(cam_eps_bf16_rounded_oracle=eps16.numpy().astype(np.float32), cam_bf16_format_err=np.array(fmt))
    np.savez_compressed(os.path.join(HERE, "g7_bf16_format.npz"), lora_seed=g7["lora_seed"], cam_seed=g7["cam_seed"], clip_seed=g7["clip_seed"],
                        uncond_seed=g7["uncond_seed"], hw=g7["hw"], t=g7["t"], **out)
    print("G7b written", flush=True)


if __name__ == "__main__":
    main()
