"""Pluecker-ray conditioning and Gaussian circle masks on device (`fmc/data/dataset.py:930-972,5365-5380`,
`train_cam_obj_ctrl.py:80-91`).

Only the functions the hot path consumes are provided; the SynFMC dataset classes (folder / CSV parsing,
captions, mask loading) are CPU data preparation and out of scope.  The reference
computes the embedding on the CPU every training step and copies 63 MB per clip to the GPU; here one
`fmc_plucker_fwd` launch writes it in HBM.
"""
from __future__ import annotations

import torch

from .. import hip_ops as K


def ray_condition(K_intr, c2w, H, W, device, flip_flag=None):
    """K_intr `[B, V, 4]`, c2w `[B, V, 4, 4]` -> `[B, V, H, W, 6]` = (o x d, d) on `device` (must be a GPU)."""
    if flip_flag is not None and bool(torch.as_tensor(flip_flag).any()):
        raise NotImplementedError("horizontal flips are never requested by the trainers (train_cam_obj_ctrl.py:87)")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("ray_condition runs on the GPU (fmc_plucker_fwd); there is no CPU path in this build")
    return K.plucker(torch.as_tensor(K_intr).to(dev), torch.as_tensor(c2w).to(dev), H, W, "bfhwc", torch.float32)


def to_plucker_embedding(c2w_rel_poses, intrinsics, sample_size, ori_h=None, ori_w=None, rescale_fxy=True,
                         device="cuda", dtype=torch.float32, layout="bfchw"):
    """`[B, F, 3, 4]` relative poses + `[B, F, 4]` intrinsics -> `[B, F, 6, H, W]` (the trainer's layout) on device.
    layout "bcfhw" gives the pose encoder's input directly, "unshuffle8" its PixelUnshuffled channels-last form."""
    intr = torch.as_tensor(intrinsics).to(device=device, dtype=torch.float32)
    c2w = torch.as_tensor(c2w_rel_poses).to(device=device, dtype=torch.float32)
    H, W = sample_size
    if layout == "bfchw":
        return K.plucker(intr, c2w, H, W, "bcfhw", dtype).permute(0, 2, 1, 3, 4)
    return K.plucker(intr, c2w, H, W, layout, dtype)


def gaussian_circle_masks(circles, H, W, device="cuda"):
    """Analytic half of the reference's sphere masks (`dataset.py:5365-5380`): `circles [..., 3]` = (cx, cy, radius)
    from `cv2.minEnclosingCircle` (host side, out of scope) -> `[..., H, W]` fp32 masks written in HBM by
    `fmc_gaussian_circle_mask_fwd`, ready for `util.get_traj_features_v2` / `fmc_omc_rasterize_fwd`."""
    return K.gaussian_circle_masks(torch.as_tensor(circles, dtype=torch.float32).to(device), H, W)
