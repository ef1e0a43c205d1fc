"""Restatement of the `diffusers==0.24.0` primitives the FMC hot path calls.

TEST INFRASTRUCTURE (see `oracle/__init__.py`).  **Parity unpinned**: diffusers
is pinned by the reference at `environment.yaml:13` but is not vendored, not
installed in this image and cannot be fetched; the reference holds no tests or
golden vectors for it.  Every class below restates the *published* 0.24.0
behaviour in plain fp32 PyTorch and cites the reference
call site that relies on it.  `tests/test_cpu_misc.py` (`test_attention_matches_sdpa` .. `test_positional_encoding_added_after_layernorm`) cross-checks
them against torch built-ins (`F.scaled_dot_product_attention`, `F.group_norm`,
`F.gelu`, closed-form DDIM).

Parameter / sub-module names equal diffusers' so that reference checkpoints
(SD-1.5 `diffusion_pytorch_model.bin`, AnimateDiff-v3, FMC stage ckpts) keep
their state-dict keys.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


# ----------------------------------------------------------------------------
# time embedding  (call sites: fmc/models/unet.py:112,115 ; unet_cam_obj.py:1165-1171)
# ----------------------------------------------------------------------------
class Timesteps(nn.Module):
    """Sinusoidal embedding; always returns fp32 (`unet.py:600-604` casts)."""

    def __init__(self, num_channels: int, flip_sin_to_cos: bool, downscale_freq_shift: float):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps: torch.Tensor) -> torch.Tensor:
        half = self.num_channels // 2
        k = torch.arange(half, dtype=torch.float32, device=timesteps.device)
        freqs = torch.exp(-math.log(10000.0) * k / (half - self.downscale_freq_shift))
        ang = timesteps[:, None].float() * freqs[None, :]
        sin, cos = torch.sin(ang), torch.cos(ang)
        emb = torch.cat([cos, sin], dim=-1) if self.flip_sin_to_cos else torch.cat([sin, cos], dim=-1)
        if self.num_channels % 2 == 1:
            emb = F.pad(emb, (0, 1))
        return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels: int, time_embed_dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample):
        return self.linear_2(self.act(self.linear_1(sample)))


# ----------------------------------------------------------------------------
# LoRA-compatible layers (0.24.0 passes `scale` positionally when PEFT is off:
# fmc/models/attention_processor.py:32,50,59-60,69)
# ----------------------------------------------------------------------------
class LoRACompatibleLinear(nn.Linear):
    def forward(self, hidden_states, scale: float = 1.0):  # noqa: D401 - no lora_layer attached in FMC
        return super().forward(hidden_states)


class LoRACompatibleConv(nn.Conv2d):
    def forward(self, hidden_states, scale: float = 1.0):
        return super().forward(hidden_states)


class LoRALinearLayer(nn.Module):
    """`up(down(x))`, bias free (`attention_processor.py:103-106`)."""

    def __init__(self, in_features, out_features, rank=4, network_alpha=None, device=None, dtype=None):
        super().__init__()
        self.down = nn.Linear(in_features, rank, bias=False, device=device, dtype=dtype)
        self.up = nn.Linear(rank, out_features, bias=False, device=device, dtype=dtype)
        self.network_alpha = network_alpha
        self.rank = rank
        self.out_features = out_features
        self.in_features = in_features
        nn.init.normal_(self.down.weight, std=1 / rank)
        nn.init.zeros_(self.up.weight)

    def forward(self, hidden_states):
        orig_dtype = hidden_states.dtype
        dtype = self.down.weight.dtype
        y = self.up(self.down(hidden_states.to(dtype)))
        if self.network_alpha is not None:
            y = y * (self.network_alpha / self.rank)
        return y.to(orig_dtype)


# ----------------------------------------------------------------------------
# ResNet / resampling  (call sites: fmc/models/unet_blocks.py:175,306,350,625)
# ----------------------------------------------------------------------------
def get_activation(name: str) -> nn.Module:
    name = name.lower()
    if name in ("swish", "silu"):
        return nn.SiLU()
    if name == "mish":
        return nn.Mish()
    if name == "gelu":
        return nn.GELU()
    if name == "relu":
        return nn.ReLU()
    raise ValueError(f"Unsupported activation function: {name}")


class ResnetBlock2D(nn.Module):
    """GN-SiLU-conv3x3, + Linear(SiLU(temb)) between conv1 and norm2, GN-SiLU-conv3x3,
    1x1 shortcut when the width changes, `(x+h)/output_scale_factor`."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512,
                 groups=32, groups_out=None, pre_norm=True, eps=1e-6, non_linearity="swish",
                 time_embedding_norm="default", output_scale_factor=1.0, use_in_shortcut=None):
        super().__init__()
        if time_embedding_norm != "default":
            raise NotImplementedError("FMC only builds time_embedding_norm='default' resnets")
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.output_scale_factor = output_scale_factor
        self.time_embedding_norm = time_embedding_norm
        groups_out = groups if groups_out is None else groups_out
        self.norm1 = nn.GroupNorm(num_groups=groups, num_channels=in_channels, eps=eps, affine=True)
        self.conv1 = LoRACompatibleConv(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.time_emb_proj = LoRACompatibleLinear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(num_groups=groups_out, num_channels=out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = LoRACompatibleConv(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.nonlinearity = get_activation(non_linearity)
        self.use_in_shortcut = in_channels != out_channels if use_in_shortcut is None else use_in_shortcut
        self.conv_shortcut = (
            LoRACompatibleConv(in_channels, out_channels, kernel_size=1, stride=1, padding=0)
            if self.use_in_shortcut else None
        )

    def forward(self, input_tensor, temb, scale: float = 1.0):
        h = self.conv1(self.nonlinearity(self.norm1(input_tensor)))
        if self.time_emb_proj is not None and temb is not None:
            h = h + self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.dropout(self.nonlinearity(self.norm2(h))))
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return (input_tensor + h) / self.output_scale_factor


class Downsample2D(nn.Module):
    """3x3 stride-2 conv (FMC always passes use_conv=True, name='op' -> attribute `conv`)."""

    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv"):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        self.use_conv, self.padding, self.name = use_conv, padding, name
        if use_conv:
            conv = LoRACompatibleConv(self.channels, self.out_channels, 3, stride=2, padding=padding)
        else:
            assert self.channels == self.out_channels
            conv = nn.AvgPool2d(kernel_size=2, stride=2)
        if name == "conv":
            self.Conv2d_0 = conv
        self.conv = conv

    def forward(self, hidden_states, scale: float = 1.0):
        assert hidden_states.shape[1] == self.channels
        if self.use_conv and self.padding == 0:
            hidden_states = F.pad(hidden_states, (0, 1, 0, 1), mode="constant", value=0)
        return self.conv(hidden_states)


class Upsample2D(nn.Module):
    """nearest x2 (or explicit size) then 3x3 conv."""

    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv"):
        super().__init__()
        assert not use_conv_transpose
        self.channels, self.out_channels = channels, out_channels or channels
        self.use_conv, self.name = use_conv, name
        conv = LoRACompatibleConv(self.channels, self.out_channels, 3, padding=1) if use_conv else None
        if name == "conv":
            self.conv = conv
        else:
            self.Conv2d_0 = conv

    def forward(self, hidden_states, output_size=None, scale: float = 1.0):
        assert hidden_states.shape[1] == self.channels
        dtype = hidden_states.dtype
        if dtype == torch.bfloat16:  # 0.24.0 round-trips bf16 through fp32 for interpolate
            hidden_states = hidden_states.to(torch.float32)
        if hidden_states.shape[0] >= 64:
            hidden_states = hidden_states.contiguous()
        if output_size is None:
            hidden_states = F.interpolate(hidden_states, scale_factor=2.0, mode="nearest")
        else:
            hidden_states = F.interpolate(hidden_states, size=output_size, mode="nearest")
        if dtype == torch.bfloat16:
            hidden_states = hidden_states.to(dtype)
        if self.use_conv:
            conv = self.conv if self.name == "conv" else self.Conv2d_0
            hidden_states = conv(hidden_states)
        return hidden_states


# ----------------------------------------------------------------------------
# Attention  (base of TemporalSelfAttention, fmc/models/motion_module.py:324;
# members touched by the processors: fmc/models/attention_processor.py)
# ----------------------------------------------------------------------------
class DefaultAttnProcessor:
    """diffusers' stock un-fused processor (what a fresh `Attention` carries under
    torch 1.13, i.e. without SDPA): baddbmm -> softmax -> bmm."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 scale: float = 1.0, **_unused):
        residual = hidden_states
        b, s_kv, _ = hidden_states.shape if encoder_hidden_states is None else encoder_hidden_states.shape
        attention_mask = attn.prepare_attention_mask(attention_mask, s_kv, b)
        q = attn.to_q(hidden_states, scale)
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        k, v = attn.to_k(ctx, scale), attn.to_v(ctx, scale)
        q, k, v = attn.head_to_batch_dim(q), attn.head_to_batch_dim(k), attn.head_to_batch_dim(v)
        probs = attn.get_attention_scores(q, k, attention_mask)
        out = attn.batch_to_head_dim(torch.bmm(probs, v))
        out = attn.to_out[1](attn.to_out[0](out, scale))
        if attn.residual_connection:
            out = out + residual
        return out / attn.rescale_output_factor


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, cross_attention_norm=None,
                 cross_attention_norm_num_groups=32, added_kv_proj_dim=None, norm_num_groups=None,
                 spatial_norm_dim=None, out_bias=True, scale_qk=True, only_cross_attention=False, eps=1e-5,
                 rescale_output_factor=1.0, residual_connection=False, _from_deprecated_attn_block=False,
                 processor=None):
        super().__init__()
        assert cross_attention_norm is None and added_kv_proj_dim is None and norm_num_groups is None
        assert spatial_norm_dim is None
        self.inner_dim = dim_head * heads
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.upcast_attention, self.upcast_softmax = upcast_attention, upcast_softmax
        self.rescale_output_factor, self.residual_connection = rescale_output_factor, residual_connection
        self.dropout = dropout
        self.scale_qk = scale_qk
        self.scale = dim_head ** -0.5 if scale_qk else 1.0
        self.heads = heads
        self.sliceable_head_dim = heads
        self.added_kv_proj_dim = added_kv_proj_dim
        self.only_cross_attention = only_cross_attention
        self.group_norm = None
        self.spatial_norm = None
        self.norm_cross = None
        self.to_q = LoRACompatibleLinear(query_dim, self.inner_dim, bias=bias)
        if not only_cross_attention:
            self.to_k = LoRACompatibleLinear(self.cross_attention_dim, self.inner_dim, bias=bias)
            self.to_v = LoRACompatibleLinear(self.cross_attention_dim, self.inner_dim, bias=bias)
        else:
            self.to_k = self.to_v = None
        self.to_out = nn.ModuleList([LoRACompatibleLinear(self.inner_dim, query_dim, bias=out_bias),
                                     nn.Dropout(dropout)])
        self.set_processor(processor if processor is not None else DefaultAttnProcessor())

    # -- processor plumbing --------------------------------------------------
    def set_processor(self, processor, _remove_lora: bool = False):
        # an nn.Module processor replaces a previous one in `_modules`
        if hasattr(self, "processor") and isinstance(self.processor, nn.Module) \
                and not isinstance(processor, nn.Module):
            self._modules.pop("processor")
        self.processor = processor

    def set_use_memory_efficient_attention_xformers(self, *a, **k):
        pass

    def set_attention_slice(self, slice_size):
        pass

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **cross_attention_kwargs)

    # -- tensor helpers ------------------------------------------------------
    def head_to_batch_dim(self, tensor, out_dim=3):
        b, s, c = tensor.shape
        h = self.heads
        tensor = tensor.reshape(b, s, h, c // h).permute(0, 2, 1, 3)
        if out_dim == 3:
            tensor = tensor.reshape(b * h, s, c // h)
        return tensor

    def batch_to_head_dim(self, tensor):
        bh, s, d = tensor.shape
        h = self.heads
        return tensor.reshape(bh // h, h, s, d).permute(0, 2, 1, 3).reshape(bh // h, s, d * h)

    def get_attention_scores(self, query, key, attention_mask=None):
        dtype = query.dtype
        if self.upcast_attention:
            query, key = query.float(), key.float()
        if attention_mask is None:
            base = torch.empty(query.shape[0], query.shape[1], key.shape[1], dtype=query.dtype, device=query.device)
            beta = 0
        else:
            base, beta = attention_mask, 1
        scores = torch.baddbmm(base, query, key.transpose(-1, -2), beta=beta, alpha=self.scale)
        if self.upcast_softmax:
            scores = scores.float()
        return scores.softmax(dim=-1).to(dtype)

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        if attention_mask is None:
            return None
        if attention_mask.shape[-1] != target_length:
            attention_mask = F.pad(attention_mask, (0, target_length), value=0.0)
        if out_dim == 3 and attention_mask.shape[0] < batch_size * self.heads:
            attention_mask = attention_mask.repeat_interleave(self.heads, dim=0)
        return attention_mask


# ----------------------------------------------------------------------------
# feed-forward / transformer blocks (call sites: motion_module.py:284 ; unet_blocks.py:323-333)
# ----------------------------------------------------------------------------
class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = LoRACompatibleLinear(dim_in, dim_out * 2)

    def forward(self, hidden_states, scale: float = 1.0):
        hidden_states, gate = self.proj(hidden_states).chunk(2, dim=-1)
        return hidden_states * F.gelu(gate)  # exact (erf) GELU


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False):
        super().__init__()
        if activation_fn != "geglu":
            raise NotImplementedError("FMC only builds geglu feed-forwards")
        inner = int(dim * mult)
        dim_out = dim if dim_out is None else dim_out
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(dropout), LoRACompatibleLinear(inner, dim_out)])

    def forward(self, hidden_states, scale: float = 1.0):
        for m in self.net:
            hidden_states = m(hidden_states)
        return hidden_states


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, num_attention_heads, attention_head_dim, dropout=0.0, cross_attention_dim=None,
                 activation_fn="geglu", attention_bias=False, only_cross_attention=False,
                 double_self_attention=False, upcast_attention=False, norm_elementwise_affine=True,
                 norm_type="layer_norm", norm_eps=1e-5, final_dropout=False):
        super().__init__()
        assert norm_type == "layer_norm"
        self.only_cross_attention = only_cross_attention
        self.norm1 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine, eps=norm_eps)
        self.attn1 = Attention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim,
                               dropout=dropout, bias=attention_bias,
                               cross_attention_dim=cross_attention_dim if only_cross_attention else None,
                               upcast_attention=upcast_attention)
        if cross_attention_dim is not None or double_self_attention:
            self.norm2 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine, eps=norm_eps)
            self.attn2 = Attention(query_dim=dim,
                                   cross_attention_dim=cross_attention_dim if not double_self_attention else None,
                                   heads=num_attention_heads, dim_head=attention_head_dim, dropout=dropout,
                                   bias=attention_bias, upcast_attention=upcast_attention)
        else:
            self.norm2 = self.attn2 = None
        self.norm3 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine, eps=norm_eps)
        self.ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn, final_dropout=final_dropout)

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, timestep=None, cross_attention_kwargs=None, class_labels=None):
        # unknown kwargs go straight through to the processor in 0.24.0 (hence the
        # mandatory `traj_features` pop at fmc/modified_modules.py:54-58)
        kw = dict(cross_attention_kwargs) if cross_attention_kwargs is not None else {}
        kw.pop("gligen", None)
        h = self.norm1(hidden_states)
        hidden_states = self.attn1(
            h, encoder_hidden_states=encoder_hidden_states if self.only_cross_attention else None,
            attention_mask=attention_mask, **kw) + hidden_states
        if self.attn2 is not None:
            h = self.norm2(hidden_states)
            hidden_states = self.attn2(h, encoder_hidden_states=encoder_hidden_states,
                                       attention_mask=encoder_attention_mask, **kw) + hidden_states
        return self.ff(self.norm3(hidden_states)) + hidden_states


class _Sample:
    """Stand-in for `Transformer2DModelOutput` (callers read `.sample`)."""

    def __init__(self, sample):
        self.sample = sample


class Transformer2DModel(nn.Module):
    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=None, out_channels=None,
                 num_layers=1, dropout=0.0, norm_num_groups=32, cross_attention_dim=None, attention_bias=False,
                 activation_fn="geglu", use_linear_projection=False, only_cross_attention=False,
                 double_self_attention=False, upcast_attention=False, norm_type="layer_norm",
                 norm_elementwise_affine=True):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        self.use_linear_projection = use_linear_projection
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(num_groups=norm_num_groups, num_channels=in_channels, eps=1e-6, affine=True)
        if use_linear_projection:
            self.proj_in = LoRACompatibleLinear(in_channels, inner)
        else:
            self.proj_in = LoRACompatibleConv(in_channels, inner, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, dropout=dropout,
                                  cross_attention_dim=cross_attention_dim, activation_fn=activation_fn,
                                  attention_bias=attention_bias, only_cross_attention=only_cross_attention,
                                  double_self_attention=double_self_attention, upcast_attention=upcast_attention,
                                  norm_type=norm_type, norm_elementwise_affine=norm_elementwise_affine)
            for _ in range(num_layers)])
        self.out_channels = in_channels if out_channels is None else out_channels
        if use_linear_projection:
            self.proj_out = LoRACompatibleLinear(inner, in_channels)
        else:
            self.proj_out = LoRACompatibleConv(inner, in_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, class_labels=None,
                cross_attention_kwargs=None, attention_mask=None, encoder_attention_mask=None,
                return_dict: bool = True):
        b, _, hh, ww = hidden_states.shape
        residual = hidden_states
        hidden_states = self.norm(hidden_states)
        if not self.use_linear_projection:
            hidden_states = self.proj_in(hidden_states)
            inner = hidden_states.shape[1]
            hidden_states = hidden_states.permute(0, 2, 3, 1).reshape(b, hh * ww, inner)
        else:
            inner = hidden_states.shape[1]
            hidden_states = hidden_states.permute(0, 2, 3, 1).reshape(b, hh * ww, inner)
            hidden_states = self.proj_in(hidden_states)
        for blk in self.transformer_blocks:
            hidden_states = blk(hidden_states, attention_mask=attention_mask,
                                encoder_hidden_states=encoder_hidden_states,
                                encoder_attention_mask=encoder_attention_mask, timestep=timestep,
                                cross_attention_kwargs=cross_attention_kwargs, class_labels=class_labels)
        if not self.use_linear_projection:
            hidden_states = hidden_states.reshape(b, hh, ww, inner).permute(0, 3, 1, 2).contiguous()
            hidden_states = self.proj_out(hidden_states)
        else:
            hidden_states = self.proj_out(hidden_states)
            hidden_states = hidden_states.reshape(b, hh, ww, inner).permute(0, 3, 1, 2).contiguous()
        out = hidden_states + residual
        return _Sample(out) if return_dict else (out,)


# ----------------------------------------------------------------------------
# DDIM  (call sites: train_cam_obj_ctrl.py:231,802 ; pipeline_animation_cm_om.py:624,705,720)
# ----------------------------------------------------------------------------
class DDIMScheduler:
    """eta = 0 DDIM with `timestep_spacing="leading"`, epsilon prediction, no clipping /
    thresholding -- the only configuration the reference instantiates
    (`configs/cam.yaml:130-136`)."""

    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 steps_offset=0, clip_sample=True, set_alpha_to_one=True, prediction_type="epsilon"):
        if beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)      # (diffusers 0.24.0 scheduling_ddim.py: torch.linspace, fp32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(beta_schedule)
        self.betas = betas
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.clip_sample = clip_sample
        self.prediction_type = prediction_type
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        ts = ts + self.steps_offset
        self.timesteps = torch.from_numpy(ts).to(device)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        timesteps = timesteps.to(original_samples.device)
        a = ac[timesteps] ** 0.5
        s = (1 - ac[timesteps]) ** 0.5
        while a.ndim < original_samples.ndim:
            a, s = a.unsqueeze(-1), s.unsqueeze(-1)
        return a * original_samples + s * noise

    def step(self, model_output, timestep, sample, eta: float = 0.0):
        assert eta == 0.0 and self.prediction_type == "epsilon" and not self.clip_sample
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        beta_t = 1 - a_t
        pred_x0 = (sample - beta_t ** 0.5 * model_output) / a_t ** 0.5
        direction = (1 - a_prev) ** 0.5 * model_output
        return _PrevSample(a_prev ** 0.5 * pred_x0 + direction, pred_x0)


class _PrevSample:
    def __init__(self, prev_sample, pred_original_sample):
        self.prev_sample = prev_sample
        self.pred_original_sample = pred_original_sample
