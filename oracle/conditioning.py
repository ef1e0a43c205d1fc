"""CPU restatement of the conditioning prep on the FMC hot path (TEST INFRASTRUCTURE).

Pinned against the reference itself: `tests/golden/make_golden.py` imports
`fmc.data.dataset.ray_condition`, `fmc.util.get_traj_features_v2` and
`fmc.data.utils.create_relative_matrix_of_cam_list` from `/root/reference` and
commits their outputs; `tests/test_oracle_golden.py` holds these functions to them.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch
from einops import rearrange


# ----------------------------------------------------------------------------
# Pluecker rays  (fmc/data/dataset.py:930-972, train_cam_obj_ctrl.py:80-91)
# ----------------------------------------------------------------------------
def ray_condition(K: torch.Tensor, c2w: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """K `[B,V,4]` = (fx, fy, cx, cy) in pixels, c2w `[B,V,4,4]` -> `[B,V,H,W,6]` = (o x d, d).

    Pixel centres at +0.5; `d = normalize((i-cx)/fx, (j-cy)/fy, 1) @ R^T`; `o = t`.
    The reference's horizontal-flip branch (`flip_flag`, dataset.py:944-953) is fed an
    all-False mask by the only caller (train_cam_obj_ctrl.py:87) and is not restated.
    """
    B, V = K.shape[:2]
    dt = c2w.dtype
    j, i = torch.meshgrid(torch.linspace(0, H - 1, H, dtype=dt), torch.linspace(0, W - 1, W, dtype=dt),
                          indexing="ij")
    i = i.reshape(1, 1, H * W).expand(B, V, H * W) + 0.5
    j = j.reshape(1, 1, H * W).expand(B, V, H * W) + 0.5
    fx, fy, cx, cy = K.to(dt).chunk(4, dim=-1)
    zs = torch.ones_like(i)
    xs = (i - cx) / fx * zs
    ys = (j - cy) / fy * zs
    d = torch.stack((xs, ys, zs), dim=-1)
    d = d / d.norm(dim=-1, keepdim=True)
    rays_d = d @ c2w[..., :3, :3].transpose(-1, -2)
    rays_o = c2w[..., :3, 3][:, :, None].expand_as(rays_d)
    rays_oxd = torch.linalg.cross(rays_o, rays_d, dim=-1)
    return torch.cat([rays_oxd, rays_d], dim=-1).reshape(B, V, H, W, 6)


def to_plucker_embedding(c2w_rel_poses, intrinsics, sample_size) -> torch.Tensor:
    """`[B,F,3,4]` relative c2w + `[B,F,4]` intrinsics -> `[B,F,6,H,W]`
    (train_cam_obj_ctrl.py:80-91; the trainer then rearranges to `b c f h w`, :833)."""
    intrinsics = torch.as_tensor(intrinsics)
    c2w = torch.as_tensor(c2w_rel_poses)
    B, n = c2w.shape[:2]
    bottom = torch.tensor([0, 0, 0, 1], dtype=c2w.dtype).view(1, 1, 1, 4).expand(B, n, 1, 4)
    c2w = torch.cat([c2w, bottom], dim=2)
    return ray_condition(intrinsics, c2w, sample_size[0], sample_size[1]).permute(0, 1, 4, 2, 3).contiguous()


# ----------------------------------------------------------------------------
# OMC rasteriser  (fmc/util.py:147-213)
# ----------------------------------------------------------------------------
def rasterize_objects(obj_info: Sequence[Sequence[np.ndarray]], obj_mask: Sequence[Sequence[torch.Tensor]],
                      dtype=torch.float32):
    """The part of `get_traj_features_v2` before the Adapter call.

    obj_info[b][f]: `[n_obj,12]` float64 ndarray; obj_mask[b][f]: `[n_obj,1,H,W]` float tensor.
    Per pixel the *last* object with mask>0 wins and writes `pose12*mask` and `mask`
    (util.py:173-183); everything is then multiplied by the mask channel once more (:201).
    Returns features `[(B F),13,H,W]` and mask `[(B F),1,H,W]`.
    """
    B, Fr = len(obj_info), len(obj_info[0])
    H, W = obj_mask[0][0].shape[-2:]
    traj = torch.zeros(B, Fr, H, W, 12, dtype=dtype)
    msk = torch.zeros(B, Fr, H, W, 1, dtype=dtype)
    for b in range(B):
        for f in range(Fr):
            m_all = obj_mask[b][f].permute(0, 2, 3, 1).to(dtype)                  # [n,H,W,1]
            info = torch.from_numpy(np.asarray(obj_info[b][f])).to(dtype)         # [n,12]
            for o in range(m_all.shape[0]):
                m = m_all[o]
                on = m[..., 0] > 0
                traj[b, f][on] = (info[o].view(1, 1, 12) * m)[on]
                msk[b, f][on] = m[on]
    feats = torch.cat([traj, msk], dim=-1) * msk
    return rearrange(feats, "b f h w c -> (b f) c h w"), rearrange(msk, "b f h w c -> (b f) c h w")


def get_traj_features(obj_info, obj_mask, omcm, dtype=torch.float32, null_clips=()) -> List[torch.Tensor]:
    """`get_traj_features_v2`: rasterise -> Adapter -> 4 x `b c f h w`.  The trainer passes
    `cfg_random_null_om=False` (train_cam_obj_ctrl.py:843); with it on, the clips the reference's coin flip drops
    (`null_clips` here) get their 13-channel FEATURES zeroed while the Adapter still receives their real mask
    (util.py:194-205) -- the null condition is `mask pyramid * Adapter(0)`, i.e. the propagated biases, not zero."""
    Fr = len(obj_info[0])
    feats, msk = rasterize_objects(obj_info, obj_mask, dtype)
    for b in null_clips:
        feats[b * Fr:(b + 1) * Fr] = 0
    return [rearrange(t, "(b f) c h w -> b c f h w", f=Fr) for t in omcm(feats, msk)]


# ----------------------------------------------------------------------------
# data-side helpers used to build synthetic inputs / fixtures
# ----------------------------------------------------------------------------
def gaussian_circle_mask(H: int, W: int, center, radius: float) -> np.ndarray:
    """Analytic part of the sphere mask (fmc/data/dataset.py:5365-5380): a filled disc of
    integer centre/radius (what `cv2.circle(..., int(cx), int(cy), int(r), 1, -1)` draws:
    pixels with dx^2+dy^2 <= r^2) times a Gaussian of sigma = radius/2 around the float
    centre, normalised by its maximum over the image."""
    yy, xx = np.ogrid[:H, :W]
    dist = np.sqrt((xx - center[0]) ** 2 + (yy - center[1]) ** 2)
    g = np.exp(-0.5 * (dist / (radius / 2.0)) ** 2)
    g = g / g.max()
    ic = (int(center[0]), int(center[1]))
    disc = ((xx - ic[0]) ** 2 + (yy - ic[1]) ** 2) <= int(radius) ** 2
    return (disc * g).astype(np.float64)


def relative_cam_poses(cam_rt: np.ndarray, scale_T: float = 1.0) -> np.ndarray:
    """fmc/data/utils.py:148-165 (`create_relative_matrix_of_cam_list`): `[F,3(4),4]` absolute
    poses -> `[F,12]` poses relative to frame 0 (`R_i^T R_0`, `R_i^T (t_0 - t_i) / scale`),
    frame 0 forced to the identity."""
    rt = np.asarray(cam_rt, dtype=np.float64)[:, :3, :]
    R0, t0 = rt[0, :, :3], rt[0, :, 3]
    out = np.zeros((rt.shape[0], 3, 4))
    for i in range(rt.shape[0]):
        R, t = rt[i, :, :3], rt[i, :, 3]
        out[i, :, 3] = (-R.T @ t + R.T @ t0) / scale_T
        out[i, :, :3] = R.T @ R0
    out[0] = np.eye(3, 4)
    return out.reshape(rt.shape[0], 12)
