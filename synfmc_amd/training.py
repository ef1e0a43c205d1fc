"""Training-step pieces of the FMC hot path.

Mirrors what `train_cam_obj_ctrl.py:782-943` (stage 3, OMC) and `train_cam_ctrl.py:540-665` (stage 2, CMC) do around
`pose_adaptor(...)`: biased timestep sampling, `add_noise`, the `sd_w * MSE + mask_w * masked-MSE` loss, gradient
clipping and the optimizer step -- plus the one exchange step of the path, the gradient all-reduce, done here by
`GradAllReducer` over RCCL (`torch.distributed`, backend "nccl" on ROCm) instead of `DistributedDataParallel`:

* gradients live in a few large flat buckets (views are installed as `p.grad`, nothing is copied);
* a bucket is all-reduced asynchronously the moment its last gradient has been accumulated, so the transfers overlap
  the rest of the backward (the U-Net activation backward is ~95 % of it and produces no parameter gradients);
* buckets are sized for xGMI (default 128 MiB: 7 point-to-point links per GPU, large messages amortise the ring
  latency; the whole OMC stage is 610 MB = 5 buckets) rather than DDP's 25 MiB NVSwitch default;
* parameters that never receive a gradient (the Adapter's level-3 blocks, 60.6 M of 152.5 M params: hence the
  reference's `find_unused_parameters=True`, train_cam_obj_ctrl.py:556) are handled by flushing unfinished buckets in
  `finish()`: their slots just stay zero.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F


def biased_timesteps(bsz: int, num_train_timesteps: int, omcm_min_step: int, min_step_prob: float, device,
                     generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """train_cam_obj_ctrl.py:793-800: with probability `min_step_prob` draw t from [omcm_min_step, T), else [0, omcm_min_step)."""
    if omcm_min_step > 0:
        t_rand = torch.rand(bsz, device=device, generator=generator)
        hi = torch.randint(omcm_min_step, num_train_timesteps, (bsz,), device=device, generator=generator)
        lo = torch.randint(0, omcm_min_step, (bsz,), device=device, generator=generator)
        return torch.where(t_rand < min_step_prob, hi, lo).long()
    return torch.randint(0, num_train_timesteps, (bsz,), device=device, generator=generator).long()


def masked_mse_loss(model_pred: torch.Tensor, target: torch.Tensor, obj_masks: Optional[torch.Tensor],
                    sd_loss_weight: float = 0.3, mask_loss_weight: float = 1.0, invert: bool = False) -> torch.Tensor:
    """`sd_w * MSE(pred, target) + mask_w * MSE(mask*pred, mask*target)` (train_cam_obj_ctrl.py:878-908).
    obj_masks: `[B, F, H, W]` union of the object masks at pixel resolution (bool / 0-1), brought to latent size with
    nearest interpolation (:897-899).  `invert=True` is stage 2's `1 - mask` (train_cam_ctrl.py:624)."""
    sd = F.mse_loss(model_pred.float(), target.float(), reduction="mean")
    if obj_masks is None:
        return sd
    b, f = obj_masks.shape[:2]
    m = obj_masks.to(model_pred.dtype).reshape(b * f, 1, *obj_masks.shape[2:])
    m = F.interpolate(m, size=model_pred.shape[-2:])
    m = m.reshape(b, f, 1, *m.shape[2:]).permute(0, 2, 1, 3, 4)
    if invert:
        m = 1 - m
    ml = F.mse_loss((m * model_pred).float(), (m * target).float(), reduction="mean")
    return mask_loss_weight * ml + sd_loss_weight * sd


class GradAllReducer:
    """Bucketed, overlapped gradient all-reduce (mean over ranks) for the trainable subset of a model."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 128 << 20, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        params = [p for p in params if p.requires_grad]
        self.buckets: List[dict] = []
        cur, cur_bytes = [], 0
        for p in reversed(params):                  # gradients become ready roughly in reverse registration order
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or cur[0].dtype != p.dtype or cur[0].device != p.device):
                self._add_bucket(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._add_bucket(cur)
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def _add_bucket(self, params):
        flat = torch.zeros(sum(p.numel() for p in params), dtype=params[0].dtype, device=params[0].device)
        off = 0
        for p in params:
            p.grad = flat[off: off + p.numel()].view_as(p)        # autograd accumulates in place into this view
            off += p.numel()
        self.buckets.append({"params": params, "flat": flat, "pending": len(params), "work": None, "launched": False})

    def _make_hook(self, bi):
        def hook(_p):
            b = self.buckets[bi]
            b["pending"] -= 1
            if b["pending"] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        if b["launched"]:
            return
        b["launched"] = True
        if self.world > 1:
            b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self) -> None:
        """Call after `loss.backward()`: flush buckets whose parameters never got a gradient, wait, average."""
        for b in self.buckets:
            self._launch(b)
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
                b["work"] = None
            if self.world > 1:
                b["flat"].mul_(1.0 / self.world)

    def zero_grad(self) -> None:
        for b in self.buckets:
            b["flat"].zero_()
            b["pending"] = len(b["params"])
            b["launched"] = False
        self._reinstall_views()      # an optimizer's zero_grad(set_to_none=True) may have dropped the views

    def _reinstall_views(self):
        for b in self.buckets:
            off = 0
            for p in b["params"]:
                view = b["flat"][off: off + p.numel()].view_as(p)
                if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                    p.grad = view
                off += p.numel()

    def parameters(self):
        return [p for b in self.buckets for p in b["params"]]


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Rank-0 -> all copy of the module state (what the DDP constructor does once)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def stage3_training_step(pose_adaptor, omcm, noise_scheduler, optimizer, reducer: Optional[GradAllReducer], latents,
                         noise, timesteps, encoder_hidden_states, plucker_embedding, traj_features_fn, obj_masks,
                         sd_loss_weight=0.3, mask_loss_weight=1.0, max_grad_norm=1.0):
    """One OMC-stage optimisation step (train_cam_obj_ctrl.py:802-943 minus data loading, VAE and CLIP).

    `traj_features_fn()` must run the (trainable) Adapter, e.g. `lambda: get_traj_features_v2(infos, masks, omcm, ...)`.
    Returns the loss value (a 0-d tensor)."""
    noisy_latents = noise_scheduler.add_noise(latents, noise, timesteps)
    traj_features = traj_features_fn()
    model_pred = pose_adaptor(noisy_latents, timesteps, encoder_hidden_states=encoder_hidden_states,
                              pose_embedding=plucker_embedding, traj_features=traj_features)
    loss = masked_mse_loss(model_pred, noise, obj_masks, sd_loss_weight, mask_loss_weight)
    loss.backward()
    if reducer is not None:
        reducer.finish()
    torch.nn.utils.clip_grad_norm_([p for p in omcm.parameters() if p.requires_grad], max_grad_norm)
    optimizer.step()
    if reducer is not None:
        reducer.zero_grad()
    else:
        optimizer.zero_grad(set_to_none=True)
    return loss.detach()


def stage2_trainable_parameters(unet, pose_encoder) -> List[torch.nn.Parameter]:
    """The CMC stage trains the camera encoder and the Camera-Adapter merge layers of the temporal attention
    processors (`qkv_merge` / `q_merge` / `kv_merge`), everything else is frozen (train_cam_ctrl.py:262-283)."""
    params = list(pose_encoder.parameters())
    params += [p for n, p in unet.named_parameters() if "_merge." in n]
    return params


def stage2_training_step(pose_adaptor, trainable: Iterable[torch.nn.Parameter], noise_scheduler, optimizer,
                         reducer: Optional[GradAllReducer], latents, noise, timesteps, encoder_hidden_states,
                         plucker_embedding, obj_masks=None, sd_loss_weight=0.3, mask_loss_weight=1.0, max_grad_norm=1.0):
    """One CMC-stage optimisation step (train_cam_ctrl.py:540-665 minus data loading, VAE and CLIP): the camera
    encoder runs inside `pose_adaptor` with gradients, the loss weights the background (`1 - object mask`, :624)."""
    noisy_latents = noise_scheduler.add_noise(latents, noise, timesteps)
    model_pred = pose_adaptor(noisy_latents, timesteps, encoder_hidden_states=encoder_hidden_states,
                              pose_embedding=plucker_embedding)
    loss = masked_mse_loss(model_pred, noise, obj_masks, sd_loss_weight, mask_loss_weight, invert=True)
    loss.backward()
    if reducer is not None:
        reducer.finish()
    torch.nn.utils.clip_grad_norm_([p for p in trainable if p.requires_grad], max_grad_norm)
    optimizer.step()
    if reducer is not None:
        reducer.zero_grad()
    else:
        optimizer.zero_grad(set_to_none=True)
    return loss.detach()
