"""`get_traj_features_v2` (`fmc/util.py:147-213`) on the gfx950 path.

Same signature.  The reference rasterises per (batch, frame, object) with boolean-index scatters in Python;
here the per-object poses and Gaussian masks are stacked once, `fmc_omc_rasterize_fwd` writes the 13-channel
feature map directly in the Adapter's input layout (PixelUnshuffle(8), channels-last), and the Adapter runs on
it.  Returns 4 feature maps, logically `b c f h w`.
"""
from __future__ import annotations

import random

import numpy as np
import torch

from . import hip_ops as K
from .models.pose_adaptor import features_to_video


def stack_object_inputs(obj_info_list_list, obj_mask_list_list, device):
    """lists `[B][F]` of `[n_obj, 12]` ndarrays / `[n_obj, 1, H, W]` tensors -> (`[BF, n, 12]`, `[BF, n, H, W]`) fp32
    device tensors.  Frames with fewer objects are padded with empty masks (never win the rasteriser)."""
    assert len(obj_info_list_list) == len(obj_mask_list_list)
    B, Fr = len(obj_info_list_list), len(obj_info_list_list[0])
    H, W = obj_mask_list_list[0][0].shape[-2:]
    n_max = max(m.shape[0] for ml in obj_mask_list_list for m in ml)
    poses = torch.zeros(B * Fr, n_max, 12, dtype=torch.float32)
    masks = torch.zeros(B * Fr, n_max, H, W, dtype=torch.float32)
    for b in range(B):
        for f in range(Fr):
            m = torch.as_tensor(obj_mask_list_list[b][f])
            n = m.shape[0]
            masks[b * Fr + f, :n] = m.reshape(n, H, W).to(torch.float32)
            poses[b * Fr + f, :n] = torch.from_numpy(np.asarray(obj_info_list_list[b][f])).to(torch.float32)
    return poses.to(device, non_blocking=True), masks.to(device, non_blocking=True)


def get_traj_features_v2(obj_info_list_list, obj_mask_list_list, omcm, cfg_random_null_om, cfg_random_null_om_ratio,
                         is_cm_condition_null_list, local_rank, dtype):
    B, Fr = len(obj_info_list_list), len(obj_info_list_list[0])
    device = torch.device("cuda", local_rank) if isinstance(local_rank, int) else torch.device(local_rank)
    poses, masks = stack_object_inputs(obj_info_list_list, obj_mask_list_list, device)
    # a 3-D mask tells `Adapter.forward` that the features are already PixelUnshuffled + channels-last; calling
    # `omcm(...)` (not `omcm.module`) keeps DistributedDataParallel's forward hooks in the loop
    feats, mask = K.omc_rasterize(poses, masks, "unshuffle8", dtype)
    if cfg_random_null_om:
        # util.py:194-199: a dropped clip loses its 13-channel FEATURES only; the Adapter still receives its real mask, so
        # the null condition is `mask pyramid * Adapter(0)` (the conv biases propagated), not zero
        fv = feats.view(B, Fr, *feats.shape[1:])
        for i in range(B):
            if not (random.random() > cfg_random_null_om_ratio):
                fv[i].zero_()
    return features_to_video(omcm(feats, mask), B)
