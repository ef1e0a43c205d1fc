"""CPU restatement of the two callers of the hot path (TEST INFRASTRUCTURE):
the DDIM denoising loop of `CameraObjCtrlPipeline.__call__`
(fmc/pipelines/pipeline_animation_cm_om.py:570-738, minus CLIP / VAE which are
outside the metric) and the stage-3 loss (train_cam_obj_ctrl.py:861-908)."""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn.functional as F
from einops import rearrange


@torch.no_grad()
def denoise(unet, scheduler, pose_encoder, text_embeddings, pose_embedding, latents, num_inference_steps=50,
            guidance_scale=7.5, traj_features: Optional[List[torch.Tensor]] = None, omcm_min_step: int = 0,
            callback=None):
    """text_embeddings: `[2B,77,C]` (uncond || cond) when guidance_scale>1 else `[B,77,C]`;
    pose_embedding `[B,6,F,H,W]`; latents `[B,4,F,h,w]` already scaled by init_noise_sigma.
    Follows pipeline_animation_cm_om.py:624-720 with `multidiff_total_steps == 1`."""
    cfg = guidance_scale > 1.0
    scheduler.set_timesteps(num_inference_steps)
    bs = pose_embedding.shape[0]
    pose_feats = [rearrange(x, "(b f) c h w -> b c f h w", b=bs) for x in pose_encoder(pose_embedding)]   # :657-660
    if cfg:
        pose_feats = [torch.cat([x, x], dim=0) for x in pose_feats]                                       # :668-669
        if traj_features is not None:                                                                      # :671-676
            traj_features = [torch.cat([torch.zeros_like(t), t], dim=0) for t in traj_features]
    for i, t in enumerate(scheduler.timesteps):
        traj = traj_features
        if traj_features is not None and omcm_min_step > 0 and t < omcm_min_step:                        # :682-685
            traj = None
        x = torch.cat([latents] * 2) if cfg else latents
        x = scheduler.scale_model_input(x, t)
        eps = unet(x, t, encoder_hidden_states=text_embeddings, pose_embedding_features=pose_feats,
                   traj_features=traj).sample.to(latents.dtype)
        if cfg:
            eps_u, eps_c = eps.chunk(2)
            eps = eps_u + guidance_scale * (eps_c - eps_u)                                                # :711-713
        latents = scheduler.step(eps, t, latents).prev_sample                                             # :720
        if callback is not None:
            callback(i, t, latents)
    return latents


def stage3_loss(model_pred, target, obj_masks, sd_loss_weight=0.3, mask_loss_weight=1.0):
    """`sd_w * MSE + mask_w * MSE(mask*pred, mask*target)` (train_cam_obj_ctrl.py:878-908).
    obj_masks: `[B,F,H,W]` bool union of the (hard) object masks at pixel resolution; it is
    brought to latent resolution with default (nearest) interpolation (:897-899)."""
    sd = F.mse_loss(model_pred.float(), target.float(), reduction="mean")
    B = obj_masks.shape[0]
    m = rearrange(obj_masks.to(model_pred.dtype), "b f h w -> (b f) 1 h w")
    m = F.interpolate(m, size=model_pred.shape[-2:])
    m = rearrange(m, "(b f) c h w -> b c f h w", b=B)
    ml = F.mse_loss((m * model_pred).float(), (m * target).float(), reduction="mean")
    return mask_loss_weight * ml + sd_loss_weight * sd


@torch.no_grad()
def denoise_plain(unet, scheduler, text_embeddings, latents, video_length, num_inference_steps=50, guidance_scale=7.5,
                  multidiff_total_steps: int = 1, multidiff_overlaps: int = 12, callback=None):
    """The plain text-to-video loop of `AnimationPipeline.__call__` (pipeline_animation_cm_om.py:315-440): no camera /
    OMC conditioning, sliding windows of `video_length` frames overlapping by `multidiff_overlaps` whose guided noise
    predictions are averaged where they overlap (:392-421).  `latents` `[B,4,F_total,h,w]` with
    `F_total = multidiff_total_steps * (video_length - multidiff_overlaps) + multidiff_overlaps` (:369)."""
    cfg = guidance_scale > 1.0
    scheduler.set_timesteps(num_inference_steps)
    single = video_length
    assert latents.shape[2] == multidiff_total_steps * (single - multidiff_overlaps) + multidiff_overlaps
    for i, t in enumerate(scheduler.timesteps):
        full = torch.zeros_like(latents)
        mask = torch.zeros_like(latents)
        preds = []
        for step in range(multidiff_total_steps):                                                        # :399-412
            s0 = step * (single - multidiff_overlaps)
            part = latents[:, :, s0:s0 + single].contiguous()
            mask[:, :, s0:s0 + single] += 1
            x = torch.cat([part] * 2) if cfg else part
            x = scheduler.scale_model_input(x, t)
            eps = unet(x, t, encoder_hidden_states=text_embeddings).sample.to(latents.dtype)
            if cfg:
                eps_u, eps_c = eps.chunk(2)
                eps = eps_u + guidance_scale * (eps_c - eps_u)
            preds.append(eps)
        for j, eps in enumerate(preds):                                                                  # :414-416
            s0 = j * (single - multidiff_overlaps)
            full[:, :, s0:s0 + single] += eps / mask[:, :, s0:s0 + single]
        latents = scheduler.step(full, t, latents).prev_sample                                           # :419
        if callback is not None:
            callback(i, t, latents)
    return latents
