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
  * CFG combine + DDIM update are one fused kernel on fp32 latents (:711-720);
  * the captured graphs are kept on the pipeline between calls: text, camera and OMC features are copied into the
    graph's static buffers per clip (and the per-clip Camera-Adapter pose terms recomputed in place), so a second clip
    of the same shape costs no warm-up and no capture.

`AnimationPipeline` (:40-440) is the plain text-to-video loop of the LoRA-only configuration (BASELINE configs[1]):
base U-Net, no pose encoder, sliding-window "multidiff" blending included.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Union

import numpy as np
import torch

from ..models.pose_adaptor import features_to_video


class AnimationPipelineOutput:
    def __init__(self, videos):
        self.videos = videos


class _GraphedUNet:
    """Captures `unet(x, t, text[, pose_feats, traj])` into a HIP graph whose inputs are static buffers owned here;
    `set_conditioning` refills them for a new clip, `__call__` replays with new latents / timestep."""

    def __init__(self, unet, latents_shape, text, pose_feats, traj_feats, dtype, cfg_shared_input: bool = False):
        self.unet = unet
        self.cfg_shared_input = bool(cfg_shared_input)      # the caller feeds `cat([latents] * 2)`: the U-Net may compute the shared prefix once
        dev = text.device
        self.x = torch.zeros(latents_shape, dtype=dtype, device=dev)
        self.t = torch.zeros((), dtype=torch.int64, device=dev)
        own = lambda v: v.detach().clone(memory_format=torch.preserve_format)
        self.text = own(text if text.dtype == dtype else text.to(dtype))   # (model dtype: the U-Net then hands THIS buffer to its cross-attention layers)
        self.pose = None if pose_feats is None else [own(p) for p in pose_feats]
        self.traj = None if traj_feats is None else [own(p) for p in traj_feats]
        self.graph = None
        self.out = None
        self._pose_terms = []

    def _call(self):
        kw = {}
        if self.pose is not None:
            kw = dict(pose_embedding_features=self.pose)
            if self.traj is not None or getattr(self.unet, "_pass_traj_none", False):   # (the CMC-only model takes no traj_features)
                kw["traj_features"] = self.traj
        if self.cfg_shared_input:
            kw["cfg_shared_input"] = True
        return self.unet(self.x, self.t, encoder_hidden_states=self.text, **kw).sample

    def capture(self):
        from .. import hip_ops as K
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._call()
        torch.cuda.current_stream().wait_stream(side)
        if getattr(self, "_cap_stream", None) is not None:     # re-capture (`set_conditioning` on a non-refillable runner): the old graph dies
            self.graph = None                                  # with its stream's stream-K workspace (~0.5 GB), not only the last one in __del__
            K.release_streamk_workspace(self._cap_stream)
        self._cap_stream = torch.cuda.Stream()
        K.prepare_streamk_workspace(self._cap_stream)      # allocated and zeroed eagerly, not inside the capture
        self.graph = torch.cuda.CUDAGraph()
        self._cap_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(self.graph, stream=self._cap_stream):
            self.out = self._call()
        torch.cuda.current_stream().wait_stream(self._cap_stream)
        # per-clip constants the processors computed during the warm-up (Camera-Adapter pose terms `s (W pose + b)`): the
        # graph reads them by address, so this runner keeps them alive and refreshes them in place for a new clip
        self._pose_terms = [(m, m.__dict__["_pose_term_cache"]) for m in self.unet.modules()
                            if m.__dict__.get("_pose_term_cache") is not None]
        # (the inner levels read the pose term through the q | k | v projection: `_PoseMerge._qkv_fold` -- its per-clip tensor is kept and refreshed too)
        self._qkv_folds = {id(m): m.__dict__["_qkv_fold_cache"] for m in self.unet.modules() if m.__dict__.get("_qkv_fold_cache") is not None}
        # ... and the text's k | v projections + fragment packs of the cross-attention layers (`Attention.text_kv`, once per clip): computed by the
        # warm-up calls from `self.text`, read by the graph by address, refreshed in place by `set_conditioning`
        self._text_kvs = [(m, m.__dict__["_text_kv"]) for m in self.unet.modules()
                          if m.__dict__.get("_text_kv") is not None and m.__dict__["_text_kv"][3] is self.text]
        # `set_conditioning` recomputes the pose terms from the `pose_view` tensors: that is only right while they ALIAS this runner's
        # static camera buffers (channels-last storage: the token view is free).  A layout that forced a private copy would replay the
        # next clip with this clip's camera term -- such a runner re-captures instead of refilling.
        mine = {p.untyped_storage().data_ptr() for p in (self.pose or [])}
        self._refillable = all(view.untyped_storage().data_ptr() in mine for _, (_, _, view) in self._pose_terms)

    def __del__(self):
        try:
            from .. import hip_ops as K
            if getattr(self, "_cap_stream", None) is not None:
                K.release_streamk_workspace(self._cap_stream)
        except Exception:
            pass

    def set_conditioning(self, text, pose_feats, traj_feats):
        """New clip, same shapes: refill the static buffers, recompute the pose terms into the tensors the graph reads."""
        from .. import hip_ops as K
        self.text.copy_(text)
        for dst, src in zip(self.pose or [], pose_feats or []):
            dst.copy_(src)
        for dst, src in zip(self.traj or [], traj_feats or []):
            dst.copy_(src)
        if not getattr(self, "_refillable", True):           # (pose terms were computed from private copies: capture again on the new buffers)
            for mod, _ in self._pose_terms:
                mod.__dict__.pop("_pose_term_cache", None)
                mod.__dict__.pop("_qkv_fold_cache", None)
            self.capture()
            return
        self._text_kvs = [(mod, mod.refresh_text_kv(entry)) for mod, entry in getattr(self, "_text_kvs", [])]
        for mod, (key, term, pose_view) in self._pose_terms:
            pf = pose_view if pose_view.is_contiguous() else pose_view.contiguous()
            term.copy_(K.linear(pf, mod.qkv_merge.weight, mod.qkv_merge.bias, None, key[-1]))
            fold = getattr(self, "_qkv_folds", {}).get(id(mod))    # the merge folded into q | k | v (inner levels): its per-clip term follows the pose term
            if fold is not None:
                assert fold[4] is term, "internal: the folded q | k | v term was built from another pose term"
                fold[2].copy_(K.linear(term, fold[3]))

    def __call__(self, x, t):
        self.x.copy_(x)
        self.t.fill_(int(t))
        self.graph.replay()
        return self.out


def _weights_version(module) -> int:
    """Fingerprint of everything a captured graph of `module` reads by address or was specialised on: identity, storage and version of
    every parameter and buffer (`p.data = ...`, `load_state_dict(assign=True)`, in-place updates), the attention-processor classes, and
    the module flags that change the launched kernels (fp8 temporal attention, LoRA / pose scales)."""
    items = [(id(t), t.data_ptr(), t._version) for t in list(module.parameters()) + list(module.buffers())]
    flags = []
    for m in module.modules():
        proc = m.__dict__.get("processor") or m._modules.get("processor") if hasattr(m, "_modules") else None
        if proc is not None:
            flags.append((type(proc).__name__, getattr(proc, "scale", None), getattr(proc, "lora_scale", None)))
        for name in ("_fp8_scales", "lora_scale", "motion_lora_scale"):
            v = m.__dict__.get(name)
            if v is not None:
                flags.append((name, id(v) if not isinstance(v, (int, float)) else v))
    return hash((tuple(items), tuple(flags)))


class AnimationPipeline:
    """Plain text-to-video loop.  Reference: pipeline_animation_cm_om.py:40-440 --
    `AnimationPipeline(vae, text_encoder, tokenizer, unet, scheduler)(prompt, video_length, height, width,
    num_inference_steps, guidance_scale, ...)`.  One denoising step = the base U-Net at CFG batch 2 on every sliding
    window (`multidiff_total_steps` windows of `video_length` frames overlapping by `multidiff_overlaps`, noise predictions
    averaged where windows overlap, :399-421) + the DDIM update."""
    vae_scale_factor = 8

    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler):
        from ..schedulers import coerce_scheduler
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.unet, self.scheduler = unet, coerce_scheduler(scheduler)       # accepts a diffusers.DDIMScheduler as well
        self._runners = {}

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

    def check_inputs(self, prompt, height, width, callback_steps):
        if prompt is not None and not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_steps is None or not isinstance(callback_steps, int) or callback_steps <= 0:
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type"
                             f" {type(callback_steps)}.")

    def prepare_latents(self, batch_size, num_channels_latents, video_length, height, width, dtype, device, generator,
                        latents=None):
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor,
                 width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective"
                             f" batch size of {batch_size}.")
        if latents is None:
            if isinstance(generator, list):
                latents = torch.cat([torch.randn((1,) + shape[1:], generator=g, device=g.device, dtype=torch.float32)
                                     .to(device) for g in generator])
            else:
                gen_dev = generator.device if isinstance(generator, torch.Generator) else torch.device(device)
                latents = torch.randn(shape, generator=generator, device=gen_dev, dtype=torch.float32).to(device)
        else:
            if latents.shape != shape:
                raise ValueError(f"Unexpected latents shape, got {latents.shape}, expected {shape}")
            latents = latents.to(device)
        return latents.float() * self.scheduler.init_noise_sigma

    # ---- shared machinery of the two loops -----------------------------------------------------------
    def _prologue(self, prompt, height, width, callback_steps, latents, num_videos_per_prompt, guidance_scale,
                  negative_prompt, prompt_embeds, device, eta):
        if eta != 0.0:
            raise NotImplementedError("eta > 0 is never used by FMC")
        unet = self.unet
        if hasattr(unet, "invalidate_text_conditioning"):
            unet.invalidate_text_conditioning()     # a new clip: no text k | v of the previous prompt survives (they are re-made once below / by the runner)
        height = height or unet.config.sample_size * self.vae_scale_factor
        width = width or unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, height, width, callback_steps)
        do_cfg = guidance_scale > 1.0
        # the loops below feed the U-Net `cat([latents] * 2)` under classifier-free guidance: its two halves are identical until the
        # first text cross-attention, and the U-Net may compute that prefix once (the `cfg_shared_input` keyword of its forward, passed per call:
        # nothing sticky on the module; FMC_CFG_SHARED=0: A/B)
        self._cfg_shared = bool(do_cfg) and os.environ.get("FMC_CFG_SHARED", "1") != "0"
        batch_size = 1
        if latents is not None:
            batch_size = latents.shape[0]
        if isinstance(prompt, list):
            batch_size = len(prompt)
        if prompt_embeds is None:
            prompt = prompt if isinstance(prompt, list) else [prompt] * batch_size
            if negative_prompt is not None and not isinstance(negative_prompt, list):
                negative_prompt = [negative_prompt] * batch_size
            prompt_embeds = self._encode_prompt(prompt, device, num_videos_per_prompt, do_cfg, negative_prompt)
        text = prompt_embeds.to(device=device, dtype=unet.dtype)
        return height, width, do_cfg, batch_size, text

    def _runner(self, x_shape, text, pose_feats, traj, use_graph):
        """The U-Net step as a callable `(x, t) -> eps`: a cached HIP graph (refilled with this clip's conditioning) or eager."""
        unet = self.unet
        shared = bool(getattr(self, "_cfg_shared", False)) and getattr(unet, "_accepts_cfg_shared_input", False)   # (a foreign U-Net never sees the keyword)
        skw = {"cfg_shared_input": True} if shared else {}
        if not use_graph:
            if hasattr(unet, "prepare_text_conditioning") and not torch.is_grad_enabled():
                unet.prepare_text_conditioning(text)     # once per clip, before the loop -- not inside step 1
            def eager(x, t):
                kw = {}
                if pose_feats is not None:
                    kw["pose_embedding_features"] = pose_feats
                    if traj is not None or getattr(unet, "_pass_traj_none", False):    # (UNet3DConditionModelPoseCond takes no traj_features)
                        kw["traj_features"] = traj
                return unet(x, torch.tensor(int(t), device=x.device), encoder_hidden_states=text, **skw, **kw).sample
            return eager
        key = (tuple(x_shape), tuple(text.shape), unet.dtype, pose_feats is not None, traj is not None,
               shared, _weights_version(unet))
        r = self._runners.get(key)
        if r is None:
            for k in [k for k in self._runners if k[:-1] == key[:-1]]:      # same shapes, stale weights: drop the graph
                del self._runners[k]
            r = _GraphedUNet(unet, x_shape, text, pose_feats, traj, unet.dtype, cfg_shared_input=shared)
            r.capture()
            self._runners[key] = r
        else:
            r.set_conditioning(text, pose_feats, traj)
        return r

    def _finish(self, latents, output_type, return_dict):
        if output_type == "latent":
            video = latents
        else:
            video = self.decode_latents(latents)
            if output_type == "tensor":
                video = torch.from_numpy(video)
        if not return_dict:
            return video
        return AnimationPipelineOutput(videos=video)

    # ---- the plain denoising loop (BASELINE configs[1]) ----------------------------------------------
    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str], None], video_length: Optional[int], height: Optional[int] = None,
                 width: Optional[int] = None, num_inference_steps: int = 50, guidance_scale: float = 7.5,
                 negative_prompt=None, num_videos_per_prompt: Optional[int] = 1, eta: float = 0.0, generator=None,
                 latents: Optional[torch.Tensor] = None, output_type: Optional[str] = "tensor",
                 return_dict: bool = True, callback: Optional[Callable] = None, callback_steps: Optional[int] = 1,
                 multidiff_total_steps: int = 1, multidiff_overlaps: int = 12, prompt_embeds=None,
                 use_graph: bool = True, **kwargs):
        unet = self.unet
        device = next(unet.parameters()).device
        height, width, do_cfg, batch_size, text = self._prologue(
            prompt, height, width, callback_steps, latents, num_videos_per_prompt, guidance_scale, negative_prompt,
            prompt_embeds, device, eta)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps = self.scheduler._timesteps_host
        single = video_length
        video_length = multidiff_total_steps * (video_length - multidiff_overlaps) + multidiff_overlaps       # :369
        latents = self.prepare_latents(batch_size * num_videos_per_prompt, unet.in_channels, video_length, height,
                                       width, text.dtype, device, generator, latents).contiguous()
        x_shape = (latents.shape[0] * (2 if do_cfg else 1), latents.shape[1], single) + tuple(latents.shape[3:])
        run = self._runner(x_shape, text, None, None, use_graph)
        stride = single - multidiff_overlaps
        for i, t in enumerate(timesteps):
            if multidiff_total_steps == 1:
                x = torch.cat([latents] * 2) if do_cfg else latents
                eps = run(x.to(unet.dtype), t)
                latents = self.scheduler.step_cfg(eps, t, latents, guidance_scale, do_cfg)
            else:                                       # sliding windows: average the guided predictions (:399-421)
                full = torch.zeros_like(latents)
                count = torch.zeros_like(latents)
                for w in range(multidiff_total_steps):
                    s0 = w * stride
                    part = latents[:, :, s0:s0 + single].contiguous()
                    x = torch.cat([part] * 2) if do_cfg else part
                    eps = run(x.to(unet.dtype), t).float()
                    if do_cfg:
                        eu, ec = eps.chunk(2)
                        eps = eu + guidance_scale * (ec - eu)
                    full[:, :, s0:s0 + single] += eps
                    count[:, :, s0:s0 + single] += 1
                latents = self.scheduler.step_cfg((full / count).contiguous(), t, latents, 1.0, False)
            if callback is not None and i % callback_steps == 0:
                callback(i, t, latents)
        return self._finish(latents, output_type, return_dict)


class CameraObjCtrlPipeline(AnimationPipeline):
    """Reference: pipeline_animation_cm_om.py:442-738 (a subclass of `AnimationPipeline` there too)."""

    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler, pose_encoder):
        super().__init__(vae, text_encoder, tokenizer, unet, scheduler)
        self.pose_encoder = pose_encoder

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
        unet = self.unet
        device = pose_embedding.device
        height, width, do_cfg, batch_size, text = self._prologue(
            prompt, height, width, callback_steps, latents, num_videos_per_prompt, guidance_scale, negative_prompt,
            prompt_embeds, device, eta)

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
        if bs != latents.shape[0] or (traj_features is not None and traj_features[0].shape[0] != latents.shape[0]):
            # the reference fails here too (`hidden_states + traj_features[idx]` / the pose merge on mismatched batches)
            raise ValueError(f"camera / OMC features cover {bs} clips but the latents hold {latents.shape[0]} "
                             "(num_videos_per_prompt > 1 needs the conditioning repeated by the caller)")
        pose_feats = features_to_video(feats, bs)
        if do_cfg:
            pose_feats = [torch.cat([x, x], dim=0) for x in pose_feats]
        pose_feats = [x.contiguous(memory_format=torch.channels_last_3d) for x in pose_feats]
        if traj_features is not None:
            traj_features = [t.to(unet.dtype).contiguous(memory_format=torch.channels_last_3d) for t in traj_features]

        omcm_min_step = kwargs.get("omcm_min_step", 0)
        x_shape = (latents.shape[0] * (2 if do_cfg else 1),) + tuple(latents.shape[1:])
        runs = {}

        for i, t in enumerate(timesteps):
            traj = traj_features
            if traj_features is not None and omcm_min_step > 0 and t < omcm_min_step:      # reference :682-685
                traj = None
            key = traj is not None
            if key not in runs:
                runs[key] = self._runner(x_shape, text, pose_feats, traj, use_graph)
            x = torch.cat([latents] * 2) if do_cfg else latents
            eps = runs[key](x.to(unet.dtype), t)
            latents = self.scheduler.step_cfg(eps, t, latents, guidance_scale, do_cfg)
            if callback is not None and i % callback_steps == 0:
                callback(i, t, latents)
        return self._finish(latents, output_type, return_dict)
