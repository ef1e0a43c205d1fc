"""DDIM scheduler for the FMC denoising loop (what the reference takes from `diffusers.DDIMScheduler`:
`train_cam_obj_ctrl.py:231,802`, `pipeline_animation_cm_om.py:624,705,720`; kwargs from
`configs/*.yaml: noise_scheduler_kwargs`).  Host side: the beta schedule and the timestep list (tiny, float64 ->
float32 like diffusers).  Device side: `step_cfg` runs the classifier-free-guidance combine and the eta=0 update
as one `fmc_cfg_ddim_step` launch on fp32 latents."""
from __future__ import annotations

import numpy as np
import torch

from . import hip_ops as K


class DDIMSchedulerOutput:
    def __init__(self, prev_sample, pred_original_sample=None):
        self.prev_sample = prev_sample
        self.pred_original_sample = pred_original_sample


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", thresholding=False, timestep_spacing="leading", **_unused):
        if trained_betas is not None:
            betas = torch.as_tensor(np.asarray(trained_betas, dtype=np.float32))
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)      # (diffusers 0.24.0 scheduling_ddim.py: torch.linspace, fp32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for {self.__class__}")
        if prediction_type != "epsilon":
            raise NotImplementedError("FMC trains epsilon prediction only (train_cam_obj_ctrl.py:870-875)")
        if clip_sample or thresholding or timestep_spacing != "leading":
            raise NotImplementedError("only clip_sample=False / leading spacing (configs/*.yaml) are built")
        self.betas = betas
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.config = type("Config", (), dict(num_train_timesteps=num_train_timesteps, steps_offset=steps_offset,
                                              prediction_type=prediction_type, clip_sample=clip_sample,
                                              beta_schedule=beta_schedule))()
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps: int, device=None):
        if num_inference_steps > self.config.num_train_timesteps:
            raise ValueError("num_inference_steps cannot exceed num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self._timesteps_host = (ts + self.config.steps_offset).tolist()
        self.timesteps = torch.tensor(self._timesteps_host, dtype=torch.int64, device=device)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _alphas(self, t: int):
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else float(self.final_alpha_cumprod)
        return a_t, a_prev

    def add_noise(self, original_samples, noise, timesteps):
        key = (original_samples.device, original_samples.dtype)
        cache = self.__dict__.setdefault("_ac_dev", {})
        ac = cache.get(key)
        if ac is None:                 # one host-to-device copy per (device, dtype): capturable in a HIP graph afterwards
            ac = cache[key] = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        timesteps = timesteps.to(original_samples.device)
        a = ac[timesteps] ** 0.5
        s = (1 - ac[timesteps]) ** 0.5
        while a.ndim < original_samples.ndim:
            a, s = a.unsqueeze(-1), s.unsqueeze(-1)
        return a * original_samples + s * noise

    def step_cfg(self, eps: torch.Tensor, timestep: int, latents: torch.Tensor, guidance_scale: float,
                 has_uncond: bool) -> torch.Tensor:
        """eps `[2B, ...]` (uncond || cond) or `[B, ...]`, latents fp32 `[B, ...]` -> new fp32 latents."""
        a_t, a_prev = self._alphas(int(timestep))
        return K.cfg_ddim_step(eps.contiguous(), latents, guidance_scale, a_t, a_prev, has_uncond)

    def step(self, model_output, timestep, sample, eta: float = 0.0, **kwargs):
        if eta != 0.0:
            raise NotImplementedError("eta > 0 is never used by FMC")
        x = sample.float().contiguous()
        out = self.step_cfg(model_output.to(torch.float32).contiguous(), int(timestep), x, 1.0, False)
        return DDIMSchedulerOutput(out.to(sample.dtype))


def coerce_scheduler(scheduler):
    """The reference's trainers hand the pipelines a `diffusers.DDIMScheduler` (train_cam_obj_ctrl.py:231, :497); the
    loops here call `step_cfg` (the fused CFG + DDIM kernel).  A foreign scheduler object is therefore re-expressed as this
    module's `DDIMScheduler` from its `.config` (same betas, offset and spacing); anything that is not a DDIM
    configuration this path implements raises here rather than producing other numbers."""
    if scheduler is None or hasattr(scheduler, "step_cfg"):
        return scheduler
    cfg = getattr(scheduler, "config", None)
    if cfg is None:
        raise TypeError(f"cannot use {type(scheduler).__name__} as the DDIM scheduler of the FMC pipelines")
    get = (lambda k, d=None: cfg.get(k, d)) if hasattr(cfg, "get") else (lambda k, d=None: getattr(cfg, k, d))
    if "DDIM" not in type(scheduler).__name__:
        raise NotImplementedError(f"{type(scheduler).__name__}: the FMC pipelines are built for DDIMScheduler (configs/*.yaml)")
    return DDIMScheduler(num_train_timesteps=get("num_train_timesteps", 1000), beta_start=get("beta_start", 0.0001),
                         beta_end=get("beta_end", 0.02), beta_schedule=get("beta_schedule", "linear"),
                         trained_betas=get("trained_betas"), clip_sample=get("clip_sample", True),
                         set_alpha_to_one=get("set_alpha_to_one", True), steps_offset=get("steps_offset", 0),
                         prediction_type=get("prediction_type", "epsilon"), thresholding=get("thresholding", False),
                         timestep_spacing=get("timestep_spacing", "leading"))
