"""Denoising pipelines of FMC (`fmc/pipelines/pipeline_animation_cm_om.py`) on the gfx950 path.

`CameraObjCtrlPipeline(vae, text_encoder, tokenizer, unet, scheduler, pose_encoder)(prompt, pose_embedding,
video_length, traj_features, height, width, num_inference_steps, guidance_scale, generator, omcm_min_step, ...)`
keeps the reference's constructor and call signature (:442-738).  The VAE and the CLIP text encoder are frozen
third-party models outside the metric: they are used through duck typing when given,
and `prompt_embeds=` / `output_type="latent"` let the loop run without them.

What changes under the hood (one denoising step = one U-Net forward at CFG batch 2 + the DDIM update):
  * camera features are computed once per clip and stay channels-last; they are NOT duplicated for the
    unconditional branch by a concat per level per step -- one `cat` per level per clip (reference :668-669);
  * OMC features are NOT zero-padded for the unconditional half (reference :671-676): `fmc_feature_add_fwd` simply
    skips that half;
  * `omcm_min_step` gating (:682-685) picks between two captured HIP graphs (with / without OMC features);
  * CFG combine + DDIM update are one fused kernel on fp32 latents (:711-720).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Union

import numpy as np
import torch

from ..models.pose_adaptor import features_to_video


class AnimationPipelineOutput:
    def __init__(self, videos):
        self.videos = videos


class _GraphedUNet:
    """Captures `unet(x, t, text, pose_feats, traj)` into a HIP graph; replays with new latents / timestep."""

    def __init__(self, unet, latents_shape, text, pose_feats, traj_feats, dtype):
        self.unet = unet
        dev = text.device
        self.x = torch.zeros(latents_shape, dtype=dtype, device=dev)
        self.t = torch.zeros((), dtype=torch.int64, device=dev)
        self.text, self.pose, self.traj = text, pose_feats, traj_feats
        self.graph = None
        self.out = None

    def _call(self):
        return self.unet(self.x, self.t, encoder_hidden_states=self.text, pose_embedding_features=self.pose,
                         traj_features=self.traj).sample

    def capture(self):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._call()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._call()
        # per-clip constants the processors computed during the warm-up (Camera-Adapter pose terms): the graph reads
        # them, so they must live as long as it does even if another clip replaces the processors' cache entries
        self._keep = [m.__dict__.get("_pose_term_cache") for m in self.unet.modules()]

    def __call__(self, x, t):
        self.x.copy_(x)
        self.t.fill_(int(t))
        self.graph.replay()
        return self.out


class CameraObjCtrlPipeline:
    vae_scale_factor = 8

    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler, pose_encoder):
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.unet, self.scheduler, self.pose_encoder = unet, scheduler, pose_encoder

    # ---- pieces outside the metric ------------------------------------------------------------------
    def _encode_prompt(self, prompt, device, num_videos_per_prompt, do_classifier_free_guidance, negative_prompt):
        if self.tokenizer is None or self.text_encoder is None:
            raise RuntimeError("no tokenizer / text_encoder attached: pass `prompt_embeds=` "
                               "([2B,77,C] = uncond || cond when guidance_scale > 1)")
        def enc(texts):
            ids = self.tokenizer(texts, padding="max_length", max_length=self.tokenizer.model_max_length,
                                 truncation=True, return_tensors="pt").input_ids.to(device)
            return self.text_encoder(ids)[0]
        text = enc(prompt).repeat_interleave(num_videos_per_prompt, dim=0)
        if not do_classifier_free_guidance:
            return text
        neg = [""] * len(prompt) if negative_prompt is None else negative_prompt
        return torch.cat([enc(neg).repeat_interleave(num_videos_per_prompt, dim=0), text])

    def decode_latents(self, latents):
        if self.vae is None:
            raise RuntimeError("no VAE attached: call with output_type='latent'")
        video_length = latents.shape[2]
        latents = 1 / 0.18215 * latents
        b = latents.shape[0]
        frames = latents.permute(0, 2, 1, 3, 4).reshape(b * video_length, *latents.shape[1:2], *latents.shape[3:])
        video = torch.cat([self.vae.decode(frames[i:i + 1]).sample for i in range(frames.shape[0])])
        video = video.reshape(b, video_length, *video.shape[1:]).permute(0, 2, 1, 3, 4)
        return ((video / 2 + 0.5).clamp(0, 1)).cpu().float().numpy()

    def prepare_latents(self, batch_size, num_channels_latents, video_length, height, width, dtype, device, generator,
                        latents=None):
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor,
                 width // self.vae_scale_factor)
        if latents is None:
            gen_dev = generator.device if isinstance(generator, torch.Generator) else torch.device(device)
            latents = torch.randn(shape, generator=generator, device=gen_dev, dtype=torch.float32).to(device)
        else:
            if latents.shape != shape:
                raise ValueError(f"Unexpected latents shape, got {latents.shape}, expected {shape}")
            latents = latents.to(device)
        return latents.float() * self.scheduler.init_noise_sigma

    # ---- the denoising loop (the metric's unit) ------------------------------------------------------
    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str], None], pose_embedding: torch.Tensor, video_length: Optional[int],
                 traj_features=None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5, negative_prompt=None,
                 num_videos_per_prompt: Optional[int] = 1, eta: float = 0.0, generator=None,
                 latents: Optional[torch.Tensor] = None, output_type: Optional[str] = "tensor",
                 return_dict: bool = True, callback: Optional[Callable] = None, callback_steps: Optional[int] = 1,
                 multidiff_total_steps: int = 1, multidiff_overlaps: int = 12, prompt_embeds=None,
                 use_graph: bool = True, pose_embedding_unshuffled: bool = False, **kwargs):
        assert multidiff_total_steps == 1                                    # reference :690
        if eta != 0.0:
            raise NotImplementedError("eta > 0 is never used by FMC")
        unet = self.unet
        height = height or unet.config.sample_size * self.vae_scale_factor
        width = width or unet.config.sample_size * self.vae_scale_factor
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        device = pose_embedding.device
        do_cfg = guidance_scale > 1.0
        batch_size = 1
        if latents is not None:
            batch_size = latents.shape[0]
        if isinstance(prompt, list):
            batch_size = len(prompt)
        if prompt_embeds is None:
            prompt = prompt if isinstance(prompt, list) else [prompt] * batch_size
            prompt_embeds = self._encode_prompt(prompt, device, num_videos_per_prompt, do_cfg, negative_prompt)
        text = prompt_embeds.to(device=device, dtype=unet.dtype)

        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps = self.scheduler._timesteps_host
        latents = self.prepare_latents(batch_size * num_videos_per_prompt, unet.in_channels, video_length, height,
                                       width, text.dtype, device, generator, latents).contiguous()

        # camera features: once per clip (reference :657-669)
        if pose_embedding_unshuffled:
            bs = pose_embedding.shape[0] // video_length
            feats = self.pose_encoder.forward_unshuffled(pose_embedding, bs)
        else:
            assert pose_embedding.ndim == 5
            bs = pose_embedding.shape[0]
            feats = self.pose_encoder(pose_embedding)
        pose_feats = features_to_video(feats, bs)
        if do_cfg:
            pose_feats = [torch.cat([x, x], dim=0) for x in pose_feats]
        pose_feats = [x.contiguous(memory_format=torch.channels_last_3d) for x in pose_feats]
        if traj_features is not None:
            traj_features = [t.to(unet.dtype).contiguous(memory_format=torch.channels_last_3d) for t in traj_features]

        omcm_min_step = kwargs.get("omcm_min_step", 0)
        x_shape = (latents.shape[0] * (2 if do_cfg else 1),) + tuple(latents.shape[1:])
        runners = {}

        def run_unet(x, t, traj):
            key = traj is not None
            if not use_graph:
                return unet(x, torch.tensor(int(t), device=device), encoder_hidden_states=text,
                            pose_embedding_features=pose_feats, traj_features=traj).sample
            if key not in runners:
                r = _GraphedUNet(unet, x_shape, text, pose_feats, traj, unet.dtype)
                r.capture()
                runners[key] = r
            return runners[key](x, t)

        for i, t in enumerate(timesteps):
            traj = traj_features
            if traj_features is not None and omcm_min_step > 0 and t < omcm_min_step:      # reference :682-685
                traj = None
            x = torch.cat([latents] * 2) if do_cfg else latents
            eps = run_unet(x.to(unet.dtype), t, traj)
            latents = self.scheduler.step_cfg(eps, t, latents, guidance_scale, do_cfg)
            if callback is not None and i % callback_steps == 0:
                callback(i, t, latents)

        if output_type == "latent":
            video = latents
        else:
            video = self.decode_latents(latents)
            if output_type == "tensor":
                video = torch.from_numpy(video)
        if not return_dict:
            return video
        return AnimationPipelineOutput(videos=video)


class AnimationPipeline(CameraObjCtrlPipeline):
    """Plain text-to-video loop (reference :40-440): no pose encoder, base U-Net."""

    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler):
        super().__init__(vae, text_encoder, tokenizer, unet, scheduler, None)
