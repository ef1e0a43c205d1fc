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
  reference's `find_unused_parameters=True`, train_cam_obj_ctrl.py:556) are found in a discovery step and dropped from
  the buckets (`grad = None`, as under DDP): they are neither shipped nor weight-decayed;
* optional bf16 compression of the buckets on the wire.
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
    """Bucketed, overlapped gradient all-reduce (mean over ranks) for the trainable subset of a model.

    * `find_unused=True` (the reference's `DDP(find_unused_parameters=True)`, train_cam_obj_ctrl.py:556): the FIRST step is
      a discovery step -- a parameter counts as used when its gradient hook fired on ANY rank (one all-reduce of a
      bitmap); unused parameters (the Adapter's level-3 blocks: 60.6 M of 152.5 M) are dropped from the buckets and get
      `grad = None`, so they are neither shipped (242 MB of zeros per step) nor touched by AdamW's weight decay --
      exactly what happens to them under DDP.
    * `compress_dtype=torch.bfloat16`: the buckets travel as bf16 (half the xGMI bytes); accumulation, clipping and the
      optimizer stay fp32.
    * `overlap=False`: nothing is launched from inside `backward()`; `finish()` reduces the buckets afterwards.  This is
      the mode for a HIP-graph-captured forward/backward (graph | all-reduce | graph, see bench.py --mode train).
    * a second `backward()` before `zero_grad()` raises instead of silently using un-reduced gradients.
    * a pruned parameter that IS reached on a later step (DDP re-detects unused parameters every iteration) is noticed by
      `finish()` -- its `.grad` is no longer None on some rank; one 1-element MAX all-reduce per step is the price -- its
      gradient is averaged over the ranks in a one-off collective for that step, and `zero_grad()` re-admits it to the buckets
      for good.  Ranks therefore never step on a rank-local gradient.  Build the optimizer over ALL trainable parameters (as the
      reference does): it skips `grad is None`; a HIP-graph-captured step needs a static set and cannot re-admit -- there the
      check fires at capture time.
    * the discovery step must run EAGERLY: under capture / replay the hooks do not fire and everything would look unused."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 128 << 20, group=None,
                 overlap: bool = True, compress_dtype: Optional[torch.dtype] = None, find_unused: bool = True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.bucket_bytes, self.overlap, self.compress_dtype = bucket_bytes, overlap, compress_dtype
        self._all = [p for p in params if p.requires_grad]
        self._discovering = find_unused
        self._fired = set()
        self._next = 0                              # first bucket not yet handed to the collective (launch order = index)
        self.unused: List[torch.nn.Parameter] = []
        self.buckets: List[dict] = []
        self._hooks = []
        self._readmit: List[torch.nn.Parameter] = []     # pruned parameters that received a gradient after all (see finish)
        self.readmitted = 0                             # how many were ever re-admitted (reported / tested)
        self._build(self._all)

    # ---- bucket construction ---------------------------------------------------------------------------
    def _build(self, params, old_grads=None):
        for h in self._hooks:
            h.remove()
        self._hooks, self.buckets = [], []
        cur, cur_bytes = [], 0
        for p in reversed(params):                  # gradients become ready roughly in reverse registration order
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > self.bucket_bytes or cur[0].dtype != p.dtype or cur[0].device != p.device):
                self._add_bucket(cur, old_grads)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._add_bucket(cur, old_grads)
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def _add_bucket(self, params, old_grads=None):
        flat = torch.zeros(sum(p.numel() for p in params), dtype=params[0].dtype, device=params[0].device)
        off = 0
        for p in params:
            view = flat[off: off + p.numel()].view_as(p)
            if old_grads is not None and id(p) in old_grads:
                view.copy_(old_grads[id(p)])
            p.grad = view                                         # autograd accumulates in place into this view
            off += p.numel()
        self.buckets.append({"params": params, "flat": flat, "pending": len(params), "work": None, "launched": False,
                             "comp": None})

    def _make_hook(self, bi):
        def hook(p):
            b = self.buckets[bi]
            if b["launched"]:
                raise RuntimeError("GradAllReducer: a gradient arrived for a bucket that was already reduced -- several "
                                   "backward() passes per step (gradient accumulation) are not supported; call zero_grad()")
            self._fired.add(id(p))
            b["pending"] -= 1
            if self.overlap and not self._discovering:
                # collectives must be issued in the same order on every rank, and a parameter may be reached on some ranks
                # only: buckets go out strictly in index order, a ready bucket waits for its predecessors
                while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
                    self._launch(self.buckets[self._next])
                    self._next += 1
        return hook

    # ---- the exchange ----------------------------------------------------------------------------------
    def _launch(self, b, async_op: bool = True):
        if b["launched"]:
            return
        b["launched"] = True
        if self.world > 1:
            buf = b["flat"]
            if self.compress_dtype is not None and buf.dtype != self.compress_dtype:
                b["comp"] = buf = b["flat"].to(self.compress_dtype)
            b["work"] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def _prune_unused(self):
        """End of the discovery step: used = hook fired on any rank."""
        if self._all[0].is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("GradAllReducer: the discovery step (first finish()) must run eagerly -- gradient hooks do not "
                               "fire under HIP-graph capture, every parameter would be classified unused")
        self._discovering = False
        used = torch.tensor([1.0 if id(p) in self._fired else 0.0 for p in self._all], device=self._all[0].device)
        if self.world > 1:
            dist.all_reduce(used, op=dist.ReduceOp.MAX, group=self.group)
        used = used.bool().tolist()
        if not any(used):
            raise RuntimeError("GradAllReducer: no gradient hook fired on any rank in the discovery step (was backward() run, "
                               "eagerly, before finish()?)")
        self.unused = [p for p, u in zip(self._all, used) if not u]
        if not self.unused:
            return
        keep = [p for p, u in zip(self._all, used) if u]
        old = {id(p): p.grad.detach().clone() for p in keep}
        for p in self.unused:
            p.grad = None
        self._build(keep, old)
        for b in self.buckets:                                    # this step's gradients are complete: nothing pending
            b["pending"] = 0

    def finish(self) -> None:
        """Call after `loss.backward()`: flush buckets whose parameters never got a gradient, wait, average."""
        if self._discovering:
            self._prune_unused()
        for b in self.buckets:
            self._launch(b)
        for b in self.buckets:
            if b["work"] is not None:
                if hasattr(b["work"], "wait"):
                    b["work"].wait()
                b["work"] = None
            if self.world > 1:
                if b["comp"] is not None:
                    b["flat"].copy_(b["comp"])
                    b["comp"] = None
                b["flat"].mul_(1.0 / self.world)
        if self.unused:
            self._reduce_late_gradients()

    def _reduce_late_gradients(self) -> None:
        """A pruned parameter has a gradient on some rank (autograd gave it a fresh rank-local `.grad`): average it over the ranks now,
        re-admit it to the buckets at the next `zero_grad()`.  Same collectives in the same order on every rank."""
        capturing = self._all[0].is_cuda and torch.cuda.is_current_stream_capturing()
        late_here = any(p.grad is not None for p in self.unused)
        if capturing:
            if late_here:
                raise RuntimeError("GradAllReducer: a parameter pruned as unused received a gradient inside a HIP-graph capture; a "
                                   "captured step needs a static parameter set -- run this step eagerly so that it is re-admitted")
            return                                               # (a captured step cannot negotiate: the check above is the guard)
        dev = self._all[0].device
        if self.world > 1:
            flag = torch.tensor([1.0 if late_here else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            late_any = bool(flag.item() > 0)
        else:
            late_any = late_here
        if not late_any:
            return
        bits = torch.tensor([1.0 if p.grad is not None else 0.0 for p in self.unused], device=dev)
        if self.world > 1:
            dist.all_reduce(bits, op=dist.ReduceOp.MAX, group=self.group)
        bits = bits.bool().tolist()
        readmit = [p for p, u in zip(self.unused, bits) if u]
        for p in readmit:
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            if self.world > 1:
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
                g.mul_(1.0 / self.world)
            p.grad = g
        self._readmit = readmit
        self.readmitted += len(readmit)

    def zero_grad(self) -> None:
        if self._readmit:                                        # rebuild the buckets with the re-admitted parameters (original order)
            ids = {id(p) for p in self._readmit} | {id(p) for b in self.buckets for p in b["params"]}
            self.unused = [p for p in self.unused if id(p) not in ids]
            self._readmit = []
            self._build([p for p in self._all if id(p) in ids])
        for p in self.unused:
            p.grad = None
        for b in self.buckets:
            b["flat"].zero_()
            b["pending"] = len(b["params"])
            b["launched"] = False
        self._next = 0
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

    def allreduce_bytes(self) -> int:
        """bytes one step puts on the wire per rank (after pruning / compression)"""
        e = torch.empty((), dtype=self.compress_dtype).element_size() if self.compress_dtype is not None else None
        return sum(b["flat"].numel() * (e or b["flat"].element_size()) for b in self.buckets)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Rank-0 -> all copy of the module state (what the DDP constructor does once).  c10d
    collectives write their output without touching the tensor's version counter, and every derived-weight cache of the
    models (fused QKV, channels-last / flipped filters, fp32 affine copies, interleaved GEGLU rows, batched temb) is keyed
    on it: the received values therefore land through `copy_`, which bumps it."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            buf = t.detach().clone()
            dist.broadcast(buf, src=src, group=group)
            t.copy_(buf)


def stage3_forward_backward(pose_adaptor, noise_scheduler, latents, noise, timesteps, encoder_hidden_states,
                            plucker_embedding, traj_features_fn, obj_masks, sd_loss_weight=0.3, mask_loss_weight=1.0):
    """add_noise -> Adapter -> U-Net -> loss -> backward (train_cam_obj_ctrl.py:802-915); returns the detached loss."""
    noisy_latents = noise_scheduler.add_noise(latents, noise, timesteps)
    traj_features = traj_features_fn()
    model_pred = pose_adaptor(noisy_latents, timesteps, encoder_hidden_states=encoder_hidden_states,
                              pose_embedding=plucker_embedding, traj_features=traj_features)
    loss = masked_mse_loss(model_pred, noise, obj_masks, sd_loss_weight, mask_loss_weight)
    loss.backward()
    return loss.detach()


def optimizer_update(trainable: Iterable[torch.nn.Parameter], optimizer, reducer: Optional["GradAllReducer"],
                     max_grad_norm: float = 1.0) -> None:
    """clip -> step -> zero (train_cam_obj_ctrl.py:917-943), on already averaged gradients."""
    params = [p for p in trainable if p.requires_grad and p.grad is not None]
    torch.nn.utils.clip_grad_norm_(params, max_grad_norm)
    optimizer.step()
    if reducer is not None:
        reducer.zero_grad()
    else:
        optimizer.zero_grad(set_to_none=True)


def stage3_training_step(pose_adaptor, omcm, noise_scheduler, optimizer, reducer: Optional[GradAllReducer], latents,
                         noise, timesteps, encoder_hidden_states, plucker_embedding, traj_features_fn, obj_masks,
                         sd_loss_weight=0.3, mask_loss_weight=1.0, max_grad_norm=1.0):
    """One OMC-stage optimisation step (train_cam_obj_ctrl.py:802-943 minus data loading, VAE and CLIP).

    `traj_features_fn()` must run the (trainable) Adapter, e.g. `lambda: get_traj_features_v2(infos, masks, omcm, ...)`.
    Returns the loss value (a 0-d tensor)."""
    loss = stage3_forward_backward(pose_adaptor, noise_scheduler, latents, noise, timesteps, encoder_hidden_states,
                                   plucker_embedding, traj_features_fn, obj_masks, sd_loss_weight, mask_loss_weight)
    if reducer is not None:
        reducer.finish()
    optimizer_update(omcm.parameters(), optimizer, reducer, max_grad_norm)
    return loss


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
    optimizer_update(trainable, optimizer, reducer, max_grad_norm)
    return loss.detach()
