"""Native building blocks of the FMC U-Net (the pieces the reference imports from `diffusers==0.24.0`).

Same class names, constructor arguments, sub-module / parameter names (so reference checkpoints keep
their state-dict keys) and logical tensor shapes as the reference call sites
(`fmc/models/unet_blocks.py:6-7`, `fmc/models/unet.py:13-20`, `fmc/models/motion_module.py:8-10`), but:

* activations are physically channels-last: a `(b f) c h w` tensor is a permuted view of `[(b f), h, w, c]`
  storage, so the token view `[(b f), h*w, c]` is free and GroupNorm / attention kernels read whole rows;
* GroupNorm(+SiLU), LayerNorm, GEGLU and attention run as hand-written gfx950 kernels through
  `libfmc_hip.so` (`synfmc_amd.hip_ops`); convolutions and projections are MIOpen / hipBLASLt calls on
  the same buffers.
"""
from __future__ import annotations

import os

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from .. import hip_ops as K


# ----------------------------------------------------------------------------
# layout helpers
# ----------------------------------------------------------------------------
def to_tokens(x: torch.Tensor) -> torch.Tensor:
    """`[N, C, h, w]` (any strides) -> contiguous `[N, h*w, C]` tokens; free for channels-last storage."""
    n, c, h, w = x.shape
    t = x.permute(0, 2, 3, 1)
    if not t.is_contiguous():
        t = t.contiguous()
    return K.carry_gn(x, t.view(n, h * w, c))


def from_tokens(t: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """contiguous `[N, h*w, C]` tokens -> logical `[N, C, h, w]` view over channels-last storage."""
    n, _, c = t.shape
    return K.carry_gn(t, t.view(n, h, w, c).permute(0, 3, 1, 2))


def f32_param(mod: nn.Module, name: str) -> torch.Tensor:
    """fp32 copy of an affine parameter (the kernels take fp32 gamma / beta), cached per version."""
    p = getattr(mod, name)
    if p.dtype == torch.float32:
        return p
    if torch.is_grad_enabled() and p.requires_grad:
        return p.float()                      # differentiable cast: trainable bf16 norm parameters still get a grad
    cache = mod.__dict__.setdefault("_f32_cache", {})
    key = (p.data_ptr(), p._version)
    hit = cache.get(name)
    if hit is None or hit[0] != key:
        hit = (key, p.detach().float())
        cache[name] = hit
    return hit[1]


class GroupNorm(nn.GroupNorm):
    """nn.GroupNorm on a `[N, C, h, w]` tensor, optionally fused with SiLU (`fmc_groupnorm_silu_fwd`)."""

    def forward(self, x: torch.Tensor, act: bool = False, x2: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`x2`: second channel block -- normalises `cat([x, x2], 1)` without building it."""
        n, c, h, w = x.shape
        y = K.groupnorm_silu(to_tokens(x), f32_param(self, "weight"), f32_param(self, "bias"), self.num_groups,
                             self.eps, act, None if x2 is None else to_tokens(x2), gn_tag=getattr(x, "_fmc_gn", None))
        return from_tokens(y, h, w)

    def skip(self, x: torch.Tensor, act: bool = False):
        """`(x for the residual connection, norm(x))` on a `[N, C, h, w]` tensor: one autograd node under a gradient (LayerNorm.skip)."""
        if NORM_SKIP and torch.is_grad_enabled() and x.requires_grad and x.is_cuda and x.dtype in (torch.bfloat16, torch.float32):
            n, c, h, w = x.shape
            xs, y = K.groupnorm_silu_skip(to_tokens(x), f32_param(self, "weight"), f32_param(self, "bias"), self.num_groups, self.eps, act)
            return from_tokens(xs, h, w), from_tokens(y, h, w)
        return x, self(x, act=act)


NORM_SKIP = os.environ.get("FMC_NORM_SKIP", "1") != "0"          # A/B switch: norm + skip connection as one autograd node (training)


class LayerNorm(nn.LayerNorm):
    def skip(self, x: torch.Tensor, pe: Optional[torch.Tensor] = None, pe_inner: int = 1, pe_frames: int = 1, defer: bool = False):
        """`(x for the residual connection, norm(x))`.  Under autograd the pair is ONE node whose backward adds the skip gradient inside
        the LayerNorm backward kernel (hip_ops.layernorm_skip); without a gradient it is `(x, self(x))`."""
        if NORM_SKIP and torch.is_grad_enabled() and x.requires_grad and x.is_cuda and x.dtype in (torch.bfloat16, torch.float32):
            return K.layernorm_skip(x if x.is_contiguous() else x.contiguous(), f32_param(self, "weight"), f32_param(self, "bias"), self.eps,
                                    pe, pe_inner, pe_frames)
        r = getattr(x, "_fmc_pending_add", None)
        if r is not None:                                # x = a vendor-arm projection whose `+ residual` was left to this norm (hip_ops.linear)
            if x.is_contiguous() and x.shape[-1] in (320, 640, 1280) and x.dtype == r.dtype:
                return K.layernorm_add(x, r, f32_param(self, "weight"), f32_param(self, "bias"), self.eps, pe, pe_inner, pe_frames)
            x = K.resolve_pending_add(x)
        return x, self(x, pe, pe_inner, pe_frames, defer=defer)

    def ln_spec(self, pe: Optional[torch.Tensor] = None, pe_inner: int = 1, pe_frames: int = 1, stats_only: bool = False):
        """What a producing GEMM needs to write this norm's output (or, `stats_only`, its rows' mean / rstd for a consumer GEMM that
        applies the norm itself) from its own epilogue (`hip_ops.linear(..., ln=...)`)."""
        return K.LnSpec(f32_param(self, "weight"), f32_param(self, "bias"), self.eps, pe, pe_inner, pe_frames, self._ln_key(pe, pe_inner, pe_frames),
                        stats_only)

    def _ln_key(self, pe, pe_inner, pe_frames):
        return (id(self), self.weight._version, self.bias._version, None if pe is None else (pe.data_ptr(), pe._version), pe_inner, pe_frames)

    def forward(self, x: torch.Tensor, pe: Optional[torch.Tensor] = None, pe_inner: int = 1,
                pe_frames: int = 1, defer: bool = False) -> torch.Tensor:
        """`defer`: the caller hands the result straight to `linear_op` / the GEGLU projection; when x's producer left the rows' statistics,
        x comes back marked "norm pending" and that GEMM applies it in its epilogue -- LayerNorm(x) is never written."""
        x = K.resolve_pending_add(x)                                   # (a lazily-added residual reaches a norm through `skip`; never dropped here)
        key = self._ln_key(pe, pe_inner, pe_frames)
        if defer and pe is None:
            stats = K.take_ln_stats(x, key)
            if stats is not None:
                return K.pending_ln(x, stats, f32_param(self, "weight"), f32_param(self, "bias"), self.eps)
        done = K.take_ln(x, key)                                        # the producer of x has written LayerNorm(x) already
        if done is not None:
            return done
        if not x.is_contiguous():
            x = x.contiguous()
        return K.layernorm(x, f32_param(self, "weight"), f32_param(self, "bias"), self.eps, pe, pe_inner, pe_frames)


PADDED_EDGE_CONVS = os.environ.get("FMC_PADDED_EDGE_CONVS", "1") != "0"      # A/B switch: conv_in / conv_out on the own kernel (else MIOpen)


class Conv2d(nn.Conv2d):
    """MIOpen conv on channels-last storage; 1x1 stride-1 convs run as a token GEMM (hipBLASLt)."""

    def forward(self, x: torch.Tensor, scale: float = 1.0, temb: Optional[torch.Tensor] = None,
                residual: Optional[torch.Tensor] = None, temb_div: int = 1, upsample: bool = False,
                x2: Optional[torch.Tensor] = None, emit_gn: bool = False) -> torch.Tensor:
        """conv(x) [+ temb[:, :, None, None]] [+ residual]; the two extras ride in the kernel epilogue when the
        gfx950 implicit-GEMM conv / GEMM is used.  `temb_div` > 1: image i uses temb row i // temb_div.
        `upsample`: conv(nearest-2x(x)) with the upsample folded into the kernel's operand addressing."""
        if self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0):
            n, c, h, w = x.shape
            assert temb is None
            y = linear_op(to_tokens(x), self.weight.view(self.out_channels, self.in_channels), self.bias,
                          None if residual is None else to_tokens(residual), 1.0, None if x2 is None else to_tokens(x2),
                          gn_hw=h * w if emit_gn else 0)
            return from_tokens(y, h, w)
        assert x2 is None, "two-source input: 1x1 convs only"
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad or
                                                  (residual is not None and residual.requires_grad))
        if (needs_grad and not upsample and self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1)
                and self.dilation == (1, 1) and self.groups == 1 and x.is_cuda and x.dtype == torch.bfloat16
                and self.weight.dtype == torch.bfloat16 and not self.weight.requires_grad
                and (self.bias is None or not self.bias.requires_grad) and (temb is None or not temb.requires_grad)
                and self.in_channels % 64 == 0 and self.out_channels % 64 == 0):
            # frozen filter, activation gradient only (training stages 2-3): forward and backward-data on the gfx950 kernel
            return K.conv3x3_frozen(x, self._weight_cl(), self.bias, temb, residual, temb_div)
        if (needs_grad and not upsample and temb is None and residual is None and self.kernel_size == (3, 3)
                and self.stride == (1, 1) and self.padding == (1, 1) and self.dilation == (1, 1) and self.groups == 1
                and x.is_cuda and x.dtype == torch.bfloat16 and self.in_channels % 64 == 0 and self.out_channels % 64 == 0
                and self.weight.requires_grad):
            # trainable filter under bf16 autocast (OMC Adapter / camera encoder): forward + backward-data on the gfx950 kernel
            return K.conv3x3_trainable(x, self.weight, self.bias)
        if (self.kernel_size == (3, 3) and self.dilation == (1, 1) and self.groups == 1 and x.is_cuda and not needs_grad
                and self.stride == (1, 1) and self.padding == (1, 1) and x.dtype == torch.bfloat16 and self.weight.dtype == torch.bfloat16
                and (self.in_channels % 64 or self.out_channels % 8) and temb is None and residual is None and not upsample and PADDED_EDGE_CONVS):
            return self.padded_conv3x3(x)
        if self.kernel_size == (3, 3) and self.dilation == (1, 1) and self.groups == 1 and x.is_cuda and not needs_grad:
            return K.conv3x3(x, self._weight_cl(), self.bias, temb, residual, self.stride, self.padding, temb_div,
                             upsample, emit_gn=emit_gn)
        if upsample:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        y = F.conv2d(x, self._weight_cl(), self.bias, self.stride, self.padding, self.dilation, self.groups)
        if temb is not None:
            y = y + (temb if temb_div == 1 else temb.repeat_interleave(temb_div, dim=0))[:, :, None, None]
        if residual is not None:
            y = y + residual
        return y

    def padded_conv3x3(self, x: torch.Tensor) -> torch.Tensor:
        """The edge convolutions (U-Net conv_in 4 -> 320 / conv_out 320 -> 4, VAE conv_in / conv_out) on the hand-written implicit-GEMM kernel:
        input channels zero-padded to a multiple of 64, output channels to a multiple of 8 (zero filters), the result sliced back -- no
        MIOpen kernel anywhere in the denoising step or the decoder.  (`unet.py:284-285,155-156` of the reference: plain nn.Conv2d.)"""
        cout, cin = self.weight.shape[:2]
        cin_p, cout_p = (cin + 63) // 64 * 64, (cout + 7) // 8 * 8
        key = (self.weight.data_ptr(), self.weight._version, None if self.bias is None else (self.bias.data_ptr(), self.bias._version))
        hit = self.__dict__.get("_padded")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                w = torch.zeros(cout_p, cin_p, 3, 3, dtype=self.weight.dtype, device=self.weight.device)
                w[:cout, :cin] = self.weight
                b = torch.zeros(cout_p, dtype=self.weight.dtype, device=self.weight.device)
                if self.bias is not None:
                    b[:cout] = self.bias
                hit = (key, w.contiguous(memory_format=torch.channels_last), b)
            self.__dict__["_padded"] = hit
        if cin_p != cin:
            xp = torch.zeros(x.shape[0], cin_p, x.shape[2], x.shape[3], dtype=x.dtype, device=x.device).contiguous(memory_format=torch.channels_last)
            xp[:, :cin] = x
            x = xp
        y = K.conv3x3(x.contiguous(memory_format=torch.channels_last), hit[1], hit[2], own_only=True)
        return y[:, :cout]

    def _weight_cl(self) -> torch.Tensor:
        """The filter in channels-last memory format.  ATen's MIOpen path otherwise re-lays-out the (up to 29 MB)
        filter on EVERY call when the input is channels-last; frozen weights are converted once and cached."""
        w = self.weight
        if w.is_contiguous(memory_format=torch.channels_last) or (torch.is_grad_enabled() and w.requires_grad):
            return w
        key = (w.data_ptr(), w._version)
        hit = self.__dict__.get("_w_cl")
        if hit is None or hit[0] != key:
            hit = (key, w.detach().contiguous(memory_format=torch.channels_last))
            self.__dict__["_w_cl"] = hit
        return hit[1]


def linear_op(x, weight, bias=None, residual=None, alpha: float = 1.0, x2=None, gn_hw: int = 0, ln=None, lazy_residual: bool = False):
    """`alpha * (x @ W^T + b) + residual`: fused gfx950 GEMM or hipBLASLt + epilogue passes (`hip_ops.linear`) for
    frozen bf16 weights on the GPU, plain autograd ops otherwise."""
    x = K.resolve_pending_add(x)
    if residual is not None:
        residual = K.resolve_pending_add(residual)
    if getattr(x, "_fmc_pending_ln", None) is not None and not (x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and not torch.is_grad_enabled()):
        x = K.resolve_pending_ln(x)                     # (never on the paths that defer a norm; kept so that a pending norm cannot be dropped)
    if x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16:
        grad = torch.is_grad_enabled()
        if not (grad and (x.requires_grad or weight.requires_grad or (residual is not None and residual.requires_grad)
                          or (x2 is not None and x2.requires_grad))):
            return K.linear(x, weight, bias, residual, alpha, x2, gn_hw=gn_hw, ln=ln, lazy_residual=lazy_residual)
        if x2 is not None:
            x, x2 = torch.cat([x, x2], dim=-1), None
        if not weight.requires_grad and (bias is None or not bias.requires_grad):   # frozen layer, activation gradient only
            return K.linear_frozen(x, weight, bias, residual, alpha)
    elif (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and K.F32_GEMM and not
          (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (residual is not None and residual.requires_grad)
                                        or (x2 is not None and x2.requires_grad)))):
        return K.linear(x, weight, bias, residual, alpha, x2)      # fp32 storage (parity mode): split-bf16 x3 on the same kernels
    if x2 is not None:
        x = torch.cat([x, x2], dim=-1)
    y = F.linear(x, weight, bias)
    if alpha != 1.0:
        y = y * alpha
    return y if residual is None else y + residual


class Linear(nn.Linear):
    def forward(self, x: torch.Tensor, scale: float = 1.0, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        return linear_op(x, self.weight, self.bias, residual)


# ----------------------------------------------------------------------------
# time embedding (fmc/models/unet.py:112,115)
# ----------------------------------------------------------------------------
class Timesteps(nn.Module):
    def __init__(self, num_channels: int, flip_sin_to_cos: bool, downscale_freq_shift: float):
        super().__init__()
        self.num_channels, self.flip_sin_to_cos = num_channels, flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps: torch.Tensor) -> torch.Tensor:
        half = self.num_channels // 2
        k = torch.arange(half, dtype=torch.float32, device=timesteps.device)
        freqs = torch.exp(-math.log(10000.0) * k / (half - self.downscale_freq_shift))
        ang = timesteps[:, None].float() * freqs[None, :]
        emb = torch.cat([torch.cos(ang), torch.sin(ang)] if self.flip_sin_to_cos
                        else [torch.sin(ang), torch.cos(ang)], dim=-1)
        return F.pad(emb, (0, 1)) if self.num_channels % 2 else emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels: int, time_embed_dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample):
        return self.linear_2(self.act(self.linear_1(sample)))


class LoRALinearLayer(nn.Module):
    """`up(down(x))` (fmc/models/attention_processor.py:103-106); merged into the base weight by
    `merge_lora_` when frozen, which is the case in FMC stages 2-3 and at inference."""

    def __init__(self, in_features, out_features, rank=4, network_alpha=None, device=None, dtype=None):
        super().__init__()
        self.down = nn.Linear(in_features, rank, bias=False, device=device, dtype=dtype)
        self.up = nn.Linear(rank, out_features, bias=False, device=device, dtype=dtype)
        self.network_alpha, self.rank = network_alpha, rank
        self.in_features, self.out_features = in_features, out_features
        nn.init.normal_(self.down.weight, std=1 / rank)
        nn.init.zeros_(self.up.weight)

    def delta_weight(self) -> torch.Tensor:
        w = self.up.weight.float() @ self.down.weight.float()
        return w * (self.network_alpha / self.rank) if self.network_alpha is not None else w

    def forward(self, hidden_states):
        y = self.up(self.down(hidden_states))
        return y * (self.network_alpha / self.rank) if self.network_alpha is not None else y


# ----------------------------------------------------------------------------
# ResNet / resampling (fmc/models/unet_blocks.py:175,306,350,625)
# ----------------------------------------------------------------------------
class ResnetBlock2D(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512,
                 groups=32, groups_out=None, pre_norm=True, eps=1e-6, non_linearity="swish",
                 time_embedding_norm="default", output_scale_factor=1.0, use_in_shortcut=None):
        super().__init__()
        if time_embedding_norm != "default" or non_linearity not in ("swish", "silu"):
            raise NotImplementedError("FMC builds default / SiLU resnets only")
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.output_scale_factor = output_scale_factor
        self.norm1 = GroupNorm(num_groups=groups, num_channels=in_channels, eps=eps, affine=True)
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.time_emb_proj = Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = GroupNorm(num_groups=groups if groups_out is None else groups_out, num_channels=out_channels,
                               eps=eps, affine=True)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.nonlinearity = nn.SiLU()
        self.use_in_shortcut = in_channels != out_channels if use_in_shortcut is None else use_in_shortcut
        self.conv_shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0) \
            if self.use_in_shortcut else None

    # set by the U-Net for one forward (inference): (`[clips, Cout]` slice of the batched projection, frames per clip)
    _t_pre = None

    def forward(self, input_tensor, temb, scale: float = 1.0, skip: Optional[torch.Tensor] = None):
        """`skip`: the up blocks' skip connection -- the block input is `cat([input_tensor, skip], 1)`
        (unet_blocks.py:683,798), read in place by the two consumers (norm1, the 1x1 shortcut) instead of being built."""
        t, div = None, 1
        if self._t_pre is not None:
            t, div = self._t_pre
            if t.shape[0] * div > input_tensor.shape[0]:               # shared CFG prefix: the block sees one half of the batch
                t = t[: input_tensor.shape[0] // div]
        elif self.time_emb_proj is not None and temb is not None:
            t = self.time_emb_proj(F.silu(temb))                       # [N, Cout], rides in conv1's epilogue
        if skip is not None and (self.conv_shortcut is None or (torch.is_grad_enabled() and (input_tensor.requires_grad or skip.requires_grad))):
            input_tensor, skip = torch.cat([input_tensor, skip], dim=1), None          # (under autograd: ONE tensor for norm1 and the shortcut)
        if skip is None and torch.is_grad_enabled() and input_tensor.requires_grad:
            # norm1 and the residual / shortcut use of the input as one autograd node: the skip gradient enters the GroupNorm backward kernel
            input_tensor, n1 = self.norm1.skip(input_tensor, act=True)
            h = self.conv1(n1, temb=t, temb_div=div)
            if self.conv_shortcut is not None:
                input_tensor = self.conv_shortcut(input_tensor)
            out = self.conv2(self.norm2(h, act=True), residual=input_tensor)
            return out if self.output_scale_factor == 1.0 else out / self.output_scale_factor
        if K.CONV_GN_FUSED and self._gn_fused_ok(input_tensor, skip, t):
            return self._forward_gn_fused(input_tensor, skip, t, div)
        # both convolutions feed a GroupNorm (norm2 here, the next module's norm behind conv2): where that norm would read its input
        # twice (the 40x64 level) the conv's epilogue emits its statistics (`emit_gn`, hip_ops.gn_emit_ok)
        h = self.conv1(self.norm1(input_tensor, act=True, x2=skip), temb=t, temb_div=div, emit_gn=True)
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor, x2=skip)
        out = self.conv2(self.norm2(h, act=True), residual=input_tensor, emit_gn=True)   # `input + h` rides in conv2's epilogue
        return out if self.output_scale_factor == 1.0 else out / self.output_scale_factor


    # ---- GroupNorm + SiLU inside the convolutions' operand path (opt-in: hip_ops.CONV_GN_FUSED) ---------------------------
    def _gn_fused_ok(self, x, skip, t) -> bool:
        n, c, h, w = x.shape
        cin = c + (skip.shape[1] if skip is not None else 0)
        if not (K.CONV_HALO and x.is_cuda and x.dtype == torch.bfloat16 and self.conv1.weight.dtype == torch.bfloat16 and not torch.is_grad_enabled()
                and self.output_scale_factor == 1.0 and c % 64 == 0 and cin == self.in_channels
                and self.out_channels % 64 == 0 and 160 % (self.out_channels // 32) == 0
                and self.norm1.num_groups == 32 and self.norm2.num_groups == 32 and (t is None or t.stride(1) == 1)):
            return False
        return (K.conv3x3_halo_supported(n, h, w, cin, c, self.out_channels, False)
                and K.conv3x3_halo_supported(n, h, w, self.out_channels, self.out_channels, self.out_channels, False))

    def _forward_gn_fused(self, x, skip, t, div):
        """`conv2(silu(norm2(conv1(silu(norm1(x))) + temb))) + shortcut(x)` (diffusers ResnetBlock2D.forward) with neither normalised tensor written:
        statistics of x (the producer's epilogue or one read) -> per-(image, channel) scale / shift -> conv1 reads the RAW x (+ skip, in place) and
        normalises it while staging its halo, leaves the statistics of its output -> conv2 likewise, `+ input` in its epilogue."""
        n, c, h, w = x.shape
        xt = to_tokens(x)
        st = None if skip is None else to_tokens(skip)
        tag = getattr(xt, "_fmc_gn", None)
        if (tag is not None and st is None and tag[1] == c and tag[0].shape[0] == n and tag[0].shape[2] == 32):
            part1 = tag[0]
            K.conv_halo_calls["stats_from_producer"] += 1
        else:
            part1 = K.groupnorm_partials(xt, 32, st)
        coef1 = K.groupnorm_coef(part1, f32_param(self.norm1, "weight"), f32_param(self.norm1, "bias"), h * w, self.in_channels, 32, self.norm1.eps)
        x_nhwc = xt.view(n, h, w, c)
        s_nhwc = None if st is None else st.view(n, h, w, st.shape[2])
        hid, part2 = K.conv3x3_halo(x_nhwc, self.conv1._weight_cl(), self.conv1.bias, t, None, div, False, s_nhwc, coef1, True, emit_gn=True)
        coef2 = K.groupnorm_coef(part2, f32_param(self.norm2, "weight"), f32_param(self.norm2, "bias"), h * w, self.out_channels, 32, self.norm2.eps)
        res = x if self.conv_shortcut is None else self.conv_shortcut(x, x2=skip)
        out, part3 = K.conv3x3_halo(hid, self.conv2._weight_cl(), self.conv2.bias, None, to_tokens(res).view(n, h, w, self.out_channels), 1, False,
                                    None, coef2, True, emit_gn=True)
        out = out.permute(0, 3, 1, 2)
        out._fmc_gn = (part3, self.out_channels)
        return out


class Downsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv"):
        super().__init__()
        assert use_conv, "FMC only builds conv downsamplers"
        self.channels, self.out_channels = channels, out_channels or channels
        self.use_conv, self.padding, self.name = use_conv, padding, name
        conv = Conv2d(self.channels, self.out_channels, 3, stride=2, padding=padding)
        if name == "conv":
            self.Conv2d_0 = conv
        self.conv = conv

    def forward(self, hidden_states, scale: float = 1.0):
        assert hidden_states.shape[1] == self.channels
        if self.padding == 0:
            hidden_states = F.pad(hidden_states, (0, 1, 0, 1), mode="constant", value=0)
        return self.conv(hidden_states)


class Upsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv"):
        super().__init__()
        assert use_conv and not use_conv_transpose and name == "conv"
        self.channels, self.out_channels = channels, out_channels or channels
        self.use_conv, self.name = use_conv, name
        self.conv = Conv2d(self.channels, self.out_channels, 3, padding=1)

    def forward(self, hidden_states, output_size=None, scale: float = 1.0):
        assert hidden_states.shape[1] == self.channels
        if not hidden_states.is_contiguous(memory_format=torch.channels_last):
            hidden_states = hidden_states.contiguous(memory_format=torch.channels_last)
        if output_size is None or tuple(output_size) == (2 * hidden_states.shape[-2], 2 * hidden_states.shape[-1]):
            return self.conv(hidden_states, upsample=True)           # the 4x larger tensor is never materialised
        else:
            hidden_states = F.interpolate(hidden_states, size=output_size, mode="nearest")
        return self.conv(hidden_states)


# ----------------------------------------------------------------------------
# attention (base of TemporalSelfAttention, fmc/models/motion_module.py:324)
# ----------------------------------------------------------------------------
TEXT_KV_ONCE = os.environ.get("FMC_TEXT_KV_ONCE", "1") != "0"      # A/B switch: the text's k | v (and fragment pack) once per clip instead of per step
text_kv_calls = {"computed": 0, "hit": 0}


class Attention(nn.Module):
    """Parameter container + helpers the processors use (SURVEY.md section 8b lists the members the reference
    processors touch).  The arithmetic lives in the processors (`synfmc_amd.models.attention_processor`)."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, cross_attention_norm=None,
                 cross_attention_norm_num_groups=32, added_kv_proj_dim=None, norm_num_groups=None,
                 spatial_norm_dim=None, out_bias=True, scale_qk=True, only_cross_attention=False, eps=1e-5,
                 rescale_output_factor=1.0, residual_connection=False, _from_deprecated_attn_block=False,
                 processor=None):
        super().__init__()
        assert cross_attention_norm is None and added_kv_proj_dim is None and norm_num_groups is None
        assert spatial_norm_dim is None and not only_cross_attention
        self.inner_dim = dim_head * heads
        self.query_dim = query_dim
        self.is_cross = cross_attention_dim is not None
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.upcast_attention, self.upcast_softmax = upcast_attention, upcast_softmax
        self.rescale_output_factor, self.residual_connection = rescale_output_factor, residual_connection
        self.scale = dim_head ** -0.5 if scale_qk else 1.0
        self.heads = heads
        self.sliceable_head_dim = heads
        self.group_norm = self.spatial_norm = self.norm_cross = None
        self.to_q = Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_v = Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(dropout)])
        from .attention_processor import AttnProcessor
        self.set_processor(processor if processor is not None else AttnProcessor())

    def set_processor(self, processor, _remove_lora: bool = False):
        if hasattr(self, "processor") and isinstance(self.processor, nn.Module) \
                and not isinstance(processor, nn.Module):
            self._modules.pop("processor")
        self.processor = processor
        # the merged projection weights and everything packed from them belong to the old processor (a freed merged tensor's address can come
        # back with version 0 under the caching allocator: the packed caches also hold their sources alive, see `_fused_tb` / `_fused_xb`)
        for name in ("_fused", "_fused_tb", "_fused_xb", "_text_kv"):
            self.__dict__.pop(name, None)

    def set_use_memory_efficient_attention_xformers(self, *a, **k):
        pass

    def set_attention_slice(self, slice_size):
        pass

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        if attention_mask is not None:
            raise NotImplementedError("attention masks are never passed on the FMC path")
        return None

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **cross_attention_kwargs)

    # ---- the text's k | v, once per clip -------------------------------------------------
    def text_kv(self, text: torch.Tensor, w_kv: torch.Tensor):
        """`(kv, frag)`: the fused k | v projection of the text embedding and -- where the fused text cross-attention block can run (C = 320 | 640,
        at most 80 tokens, bf16) -- its MFMA-fragment pack.  The reference projects `attn.to_k / to_v(encoder_hidden_states)` in every denoising step
        (fmc/models/attention_processor.py:58-59, :145-146); the text and the (frozen) weights are constant over the steps of a clip, so the pair is
        computed on the first call and kept while `text` and `w_kv` are the same storage at the same version (the entry holds both alive).  Under
        autograd, or while a stream is being captured, nothing is cached."""
        def compute():
            kv = linear_op(text, w_kv)
            C = w_kv.shape[0] // 2
            pack = (kv.is_cuda and kv.dtype == torch.bfloat16 and kv.ndim == 3 and C in (320, 640) and kv.shape[1] <= 80 and self.heads == 8
                    and kv.is_contiguous())
            return kv, (K.xattn_pack_kv(kv) if pack else None)
        if not TEXT_KV_ONCE or torch.is_grad_enabled() or not text.is_cuda:
            return linear_op(text, w_kv), None
        key = (text.data_ptr(), text._version, tuple(text.shape), text.dtype, w_kv.data_ptr(), w_kv._version)
        hit = self.__dict__.get("_text_kv")
        if hit is not None and hit[0] == key:
            text_kv_calls["hit"] += 1
            return hit[1], hit[2]
        kv, frag = compute()
        text_kv_calls["computed"] += 1
        if not torch.cuda.is_current_stream_capturing():      # (a capture's allocations belong to the graph's pool: not ours to keep)
            self.__dict__["_text_kv"] = (key, kv, frag, text, w_kv)
        return kv, frag

    def refresh_text_kv(self, entry=None):
        """Recompute an entry of `text_kv` IN PLACE after its text buffer was overwritten with a new clip's embedding (a captured HIP graph reads
        `kv` / `frag` by address: `_GraphedUNet.set_conditioning`), and make it this module's current entry again."""
        entry = entry if entry is not None else self.__dict__.get("_text_kv")
        if entry is None:
            return None
        _, kv, frag, text, w_kv = entry
        with torch.no_grad():
            kv.copy_(linear_op(text, w_kv))
            if frag is not None:
                frag.copy_(K.xattn_pack_kv(kv))
        key = (text.data_ptr(), text._version, tuple(text.shape), text.dtype, w_kv.data_ptr(), w_kv._version)
        entry = (key, kv, frag, text, w_kv)
        self.__dict__["_text_kv"] = entry
        return entry

    # ---- fused projection weights (cached; rebuilt when any source parameter changes) ----------------
    def fused_weights(self, lora=None, lora_scale: float = 1.0):
        """(W_qkv `[3C, Cin]` for self attention | (W_q, W_kv) for cross attention, W_out, b_out) with an
        optional frozen LoRA (`W + s * up @ down`) merged in."""
        srcs = [self.to_q.weight, self.to_k.weight, self.to_v.weight, self.to_out[0].weight]
        if torch.is_grad_enabled() and any(p.requires_grad for p in srcs):
            # trainable projections (camera encoder, CMC stage): build the fused weight inside autograd, uncached
            assert lora is None, "a LoRA on trainable base weights is not a configuration the reference uses"
            if self.is_cross:
                return self.to_q.weight, torch.cat([self.to_k.weight, self.to_v.weight], dim=0), self.to_out[0].weight
            return torch.cat([self.to_q.weight, self.to_k.weight, self.to_v.weight], dim=0), None, self.to_out[0].weight
        if lora is not None:
            for n in ("to_q_lora", "to_k_lora", "to_v_lora", "to_out_lora"):
                srcs += [getattr(lora, n).down.weight, getattr(lora, n).up.weight]
        key = tuple((p.data_ptr(), p._version) for p in srcs) + (lora_scale,)
        hit = self.__dict__.get("_fused")
        if hit is not None and hit[0] == key:
            return hit[1]
        with torch.no_grad():
            def w(lin, name):
                if lora is None:
                    return lin.weight
                return (lin.weight.float() + lora_scale * getattr(lora, name).delta_weight()).to(lin.weight.dtype)
            wq, wk, wv = w(self.to_q, "to_q_lora"), w(self.to_k, "to_k_lora"), w(self.to_v, "to_v_lora")
            wo = w(self.to_out[0], "to_out_lora")
            if self.is_cross:
                fused = (wq.contiguous(), torch.cat([wk, wv], dim=0).contiguous(), wo.contiguous())
            else:
                fused = (torch.cat([wq, wk, wv], dim=0).contiguous(), None, wo.contiguous())
        self.__dict__["_fused"] = (key, fused)
        return fused


def interleave_geglu(weight: torch.Tensor, bias: Optional[torch.Tensor], block: int = 32):
    """Row order `fmc_linear_bf16(epilogue=GEGLU)` expects: per 2 * block rows, `block` value rows then their `block` gate rows.
    block = 32 (the default: any tile width that is a multiple of 64 holds matching value / gate columns) for every arm but
    16, whose 16-wide MFMA blocks want [8 value | 8 gate] (block = 8)."""
    two_cff, k = weight.shape
    cff = two_cff // 2
    assert cff % block == 0
    w = weight.view(2, cff // block, block, k).permute(1, 0, 2, 3).reshape(two_cff, k).contiguous()
    b = None if bias is None else bias.view(2, cff // block, block).permute(1, 0, 2).reshape(two_cff).contiguous()
    return w, b


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = Linear(dim_in, dim_out * 2)

    def interleaved(self):
        """(W, b in 32-row value / gate blocks, W, b in [8 value | 8 gate] blocks or None, None): the row orders the fused kernels want, cached."""
        w = self.proj.weight
        key = (w.data_ptr(), w._version)
        hit = self.__dict__.get("_il")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                b = None if self.proj.bias is None else self.proj.bias.detach()
                il160 = interleave_geglu(w.detach(), b, 8) if (w.shape[0] // 2) % 160 == 0 else (None, None)
                hit = (key, interleave_geglu(w.detach(), b) + il160)
            self.__dict__["_il"] = hit
        return hit[1]

    def forward(self, hidden_states, scale: float = 1.0):
        w = self.proj.weight
        if hidden_states.is_cuda and (hidden_states.dtype == torch.bfloat16 or (hidden_states.dtype == torch.float32 and K.F32_GEMM)) \
                and hidden_states.dtype == w.dtype and w.shape[0] % 256 == 0 \
                and not (torch.is_grad_enabled() and (w.requires_grad or hidden_states.requires_grad)):
            il = self.interleaved()
            return K.geglu_linear(hidden_states, w, self.proj.bias, il[0], il[1], il[2], il[3])
        return K.geglu(self.proj(K.resolve_pending_ln(hidden_states)))


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False):
        super().__init__()
        if activation_fn != "geglu":
            raise NotImplementedError("FMC only builds geglu feed-forwards")
        inner = int(dim * mult)
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(dropout),
                                  Linear(inner, dim if dim_out is None else dim_out)])

    def _tail_weights(self, wp, bp):
        """The output projection folded into the transformer's proj_out (`hip_ops.fold_ff_tail`), cached per weight version."""
        out = self.net[2]
        ts = [out.weight, out.bias, wp, bp]
        key = tuple((t.data_ptr(), t._version) if t is not None else None for t in ts)
        hit = self.__dict__.get("_tail_fold")
        if hit is None or hit[0] != key:
            hit = (key, K.fold_ff_tail(out.weight, out.bias, wp, bp), ts)      # (the entry keeps the tensors alive: an equal pointer means the same storage)
            self.__dict__["_tail_fold"] = hit
        return hit[1]

    def forward_ln(self, h, norm, residual, tail=None):
        """`ff(norm(h)) + residual` with LayerNorm + GEGLU projection as one launch where `hip_ops.geglu_ln_direct` exists (the 20x32 level), else None.
        `tail = (W_p, b_p, x, gn_hw)`: the caller's next op is `proj_out(.) + x` -- where `hip_ops.ff_tail_ok` holds the result is THAT (tagged `_fmc_tail`)."""
        proj, out = self.net[0], self.net[2]
        w = proj.proj.weight
        if not K.geglu_ln_direct_ok(h, w) or getattr(h, "_fmc_pending_add", None) is not None or getattr(h, "_fmc_ln", None) is not None \
                or getattr(h, "_fmc_pending_ln", None) is not None:
            return None
        M = h.numel() // h.shape[-1]
        pipe = K.geglu_ln_pipe_ok(h, w) and (h.shape[-1] == 320 or K.GEGLU_PIPE_640)
        var = K.geglu_pipe_variant(M, h.shape[-1]) if pipe else -1
        key = (w.data_ptr(), w._version, var)
        hit = self.__dict__.get("_geglu_frag")
        if hit is None or hit[0] != key:
            hit = (key, K.pack_geglu_frag(w, 16 if var == 1 else 32) if pipe else K.pack_geglu_frag80(w))
            self.__dict__["_geglu_frag"] = hit
        blocked = K.geglu_direct_blocked_ok(h, out.weight, residual)
        if pipe:      # the gate in the shadow of the next chunk's MFMAs (csrc/geglu_pipe.hip, round 6)
            mid = K.geglu_ln_pipe(h, f32_param(norm, "weight"), f32_param(norm, "bias"), norm.eps, hit[1], proj.proj.bias, w.shape[0] // 2, blocked=blocked,
                                  variant=var)
        else:
            mid = K.geglu_ln_direct(h, f32_param(norm, "weight"), f32_param(norm, "bias"), norm.eps, hit[1], proj.proj.bias, w.shape[0] // 2, blocked=blocked)
        if blocked and tail is not None and residual is h and K.ff_tail_ok(h, out.weight, tail[0], tail[2]):
            wc, bc = self._tail_weights(tail[0], tail[1])
            y = K.ff_tail(mid, h, wc, bc, tail[2], tail[3])
            y._fmc_tail = True
            return y
        if blocked:
            return K.linear_from_blocked(mid, out.weight, out.bias, residual)
        return out(mid, residual=residual)

    def forward(self, hidden_states, scale: float = 1.0, residual: Optional[torch.Tensor] = None):
        proj, out = self.net[0], self.net[2]
        il = proj.interleaved() if (hidden_states.is_cuda and hidden_states.dtype == torch.bfloat16 and not torch.is_grad_enabled()) else None
        if il is not None and K.ff_blocked_ok(hidden_states, il[2], out.weight, residual):
            # the [M, 4C] intermediate stays tile-major between the two GEMMs: contiguous 10-KiB operand blocks for the second one
            mid = K.geglu_linear_blocked(hidden_states, il[2], il[3])
            return K.linear_from_blocked(mid, out.weight, out.bias, residual)
        return out(proj(hidden_states), residual=residual)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, num_attention_heads, attention_head_dim, dropout=0.0, cross_attention_dim=None,
                 activation_fn="geglu", attention_bias=False, only_cross_attention=False,
                 double_self_attention=False, upcast_attention=False, norm_elementwise_affine=True,
                 norm_type="layer_norm", norm_eps=1e-5, final_dropout=False):
        super().__init__()
        assert norm_type == "layer_norm" and not only_cross_attention and not double_self_attention
        self.norm1 = LayerNorm(dim, elementwise_affine=norm_elementwise_affine, eps=norm_eps)
        self.attn1 = Attention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim, dropout=dropout,
                               bias=attention_bias, upcast_attention=upcast_attention)
        if cross_attention_dim is not None:
            self.norm2 = LayerNorm(dim, elementwise_affine=norm_elementwise_affine, eps=norm_eps)
            self.attn2 = Attention(query_dim=dim, cross_attention_dim=cross_attention_dim, heads=num_attention_heads,
                                   dim_head=attention_head_dim, dropout=dropout, bias=attention_bias,
                                   upcast_attention=upcast_attention)
        else:
            self.norm2 = self.attn2 = None
        self.norm3 = LayerNorm(dim, elementwise_affine=norm_elementwise_affine, eps=norm_eps)
        self.ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn, final_dropout=final_dropout)

    def _attn2_weights(self, kw):
        """(W_q, W_kv, W_out) of the text cross-attention with a frozen LoRA merged in at the scale the processor call would resolve."""
        from .attention_processor import LoRAAttnProcessor, _require_frozen, resolve_lora_scale
        attn, proc = self.attn2, self.attn2.processor
        lora, lora_scale = None, 1.0
        if isinstance(proc, LoRAAttnProcessor):
            _require_frozen(proc)
            lora = proc
            lora_scale = resolve_lora_scale(proc, kw["scale"]) if "scale" in kw else resolve_lora_scale(proc)
        return attn.fused_weights(lora, lora_scale)

    def prepare_text(self, text: torch.Tensor, cross_attention_kwargs=None) -> bool:
        """The per-clip part of this block's text cross-attention -- k | v of the text and their fragment pack -- computed now (`UNet.prepare_text_conditioning`:
        the pipelines' eager path and `bench.py` call it before the denoising loop; a step that finds nothing prepared computes and keeps it itself)."""
        from .attention_processor import AttnProcessor, LoRAAttnProcessor
        if self.attn2 is None or type(self.attn2.processor) not in (AttnProcessor, LoRAAttnProcessor) or torch.is_grad_enabled():
            return False
        kw = dict(cross_attention_kwargs) if cross_attention_kwargs is not None else {}
        _, w_kv, _ = self._attn2_weights(kw)
        self.attn2.text_kv(text, w_kv)
        return True

    def _xattn_fused(self, hidden_states, encoder_hidden_states, kw):
        """`attn2(norm2(h), text) + h` as one launch (the text's k | v projection and its fragment pack come from `Attention.text_kv`: once per clip)."""
        attn = self.attn2
        w_q, w_kv, w_o = self._attn2_weights(kw)
        key = (w_q.data_ptr(), w_q._version, w_o.data_ptr(), w_o._version)
        hit = attn.__dict__.get("_fused_xb")
        if hit is None or hit[0] != key:
            # (the entry keeps w_q / w_o alive: an equal data pointer then means the same storage, not a recycled address)
            hit = ((key, K.pack_w_frag80(w_q), K.pack_w_frag80(w_o), (w_q, w_o)) if w_q.shape[0] == 640
                   else (key, K.pack_xattn_q40(w_q, attn.heads), K._w_tilemajor(w_o), (w_q, w_o)))
            attn.__dict__["_fused_xb"] = hit
        g, b = f32_param(self.norm2, "weight"), f32_param(self.norm2, "bias")
        ck = (g.data_ptr(), g._version, b.data_ptr(), b._version)
        c = self.__dict__.get("_fused_xc")
        if c is None or c[0] != ck:
            with torch.no_grad():
                c = (ck, g.contiguous(), b[None, :].expand(16, -1).float().contiguous())
            self.__dict__["_fused_xc"] = c
        h = hidden_states if hidden_states.is_contiguous() else hidden_states.contiguous()
        kv, frag = attn.text_kv(encoder_hidden_states, w_kv)           # (once per clip, not per step)
        per_text = h.shape[0] // encoder_hidden_states.shape[0]
        if h.shape[2] == 640:
            return K.xattn_block(h, c[1], c[2], self.norm2.eps, hit[1], kv, hit[2], attn.to_out[0].bias, attn.scale, per_text, frag=frag)
        # the 40x64 level: the feed-forward's norm3 is applied by its GEGLU projection from the row statistics this launch leaves
        if K.geglu_ln_direct_ok(h, self.ff.net[0].proj.weight):          # (the feed-forward normalises its input itself: no statistics to leave)
            return K.xattn_block(h, c[1], c[2], self.norm2.eps, hit[1], kv, hit[2], attn.to_out[0].bias, attn.scale, per_text, frag=frag)
        out, stats = K.xattn_block(h, c[1], c[2], self.norm2.eps, hit[1], kv, hit[2], attn.to_out[0].bias, attn.scale, per_text, stats_eps=self.norm3.eps,
                                   frag=frag)
        K.ln_epilogue_calls["emitted"] += 1
        out._fmc_ln = (stats, self.norm3._ln_key(None, 1, 1), True)
        return out

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, timestep=None, cross_attention_kwargs=None, class_labels=None,
                cfg_expand: bool = False, tail=None):
        """`tail`: see `FeedForward.forward_ln` (the enclosing transformer's proj_out, taken into the feed-forward's last launch where that exists)."""
        kw = dict(cross_attention_kwargs) if cross_attention_kwargs is not None else {}
        kw.pop("gligen", None)
        # the LayerNorm behind each attention leaves its output projection's epilogue where the tile holds whole rows (hip_ops.linear_ln)
        # ... as its rows' statistics only: every norm of this block feeds a GEMM (QKV / to_q / the GEGLU projection), which applies it in its
        # own epilogue on gamma-scaled weights (hip_ops.linear_lnc, `defer=True` below) -- LayerNorm(x) is neither written nor read.
        from .attention_processor import AttnProcessor, LoRAAttnProcessor

        def plain(attn):            # only these processors hand the normed tokens to `linear_op` and nowhere else (a pose merge also uses
            return type(attn.processor) in (AttnProcessor, LoRAAttnProcessor)      # them as a residual: it needs the materialised norm)
        d1, d2 = plain(self.attn1), self.attn2 is not None and plain(self.attn2)
        # the text cross-attention block of the 40x64 / 20x32 levels as ONE launch (LayerNorm + to_q + attention over the text tokens + to_out + residual:
        # hip_ops.xattn_block): plain / frozen-LoRA processor, 8 heads, tokens [images, hw, 320 | 640] with hw % 160 | 80 == 0, no mask
        fuse2 = (d2 and not cfg_expand and encoder_hidden_states is not None and encoder_attention_mask is None and hidden_states.ndim == 3
                 and encoder_hidden_states.ndim == 3 and hidden_states.shape[0] % encoder_hidden_states.shape[0] == 0
                 and K.xattn_block_supported(hidden_states, encoder_hidden_states.shape[1], self.attn2.heads)
                 and self.attn2.inner_dim == hidden_states.shape[2] and not self.attn2.residual_connection and self.attn2.rescale_output_factor == 1.0
                 and self.attn2.to_q.weight.dtype == torch.bfloat16 and encoder_hidden_states.dtype == torch.bfloat16)
        # (`_lazy_res`: the output projection's `+ residual` may be left to the next norm's pass when the projection runs on the vendor arm;
        #  true exactly where the very next consumer is `norm.skip` below)
        self.attn1.__dict__["_lazy_res"] = not cfg_expand and not torch.is_grad_enabled()
        if self.attn2 is not None:
            self.attn2.__dict__["_lazy_res"] = not torch.is_grad_enabled()
        if fuse2:                                   # (the fused block normalises its input itself and wants it materialised)
            self.attn1.__dict__["_lazy_res"] = False
            self.attn1.__dict__["_next_ln"] = None
        ff_direct = not torch.is_grad_enabled() and hidden_states.ndim == 3 and K.geglu_ln_direct_ok(hidden_states, self.ff.net[0].proj.weight)
        if fuse2:
            pass
        elif self.attn2 is not None:
            self.attn1.__dict__["_next_ln"] = None if (cfg_expand or torch.is_grad_enabled()) else self.norm2.ln_spec(stats_only=d2)
            self.attn2.__dict__["_next_ln"] = None if (torch.is_grad_enabled() or ff_direct) else self.norm3.ln_spec(stats_only=True)
            if ff_direct:
                self.attn2.__dict__["_lazy_res"] = False
        else:
            self.attn1.__dict__["_next_ln"] = None if (cfg_expand or torch.is_grad_enabled() or ff_direct) else self.norm3.ln_spec(stats_only=True)
        # `attn(...) + hidden_states` / `ff(...) + hidden_states`: the residual rides in the output projection's epilogue
        # (`norm.skip`: under autograd, the norm and the residual use of its input are one node -- hip_ops.layernorm_skip)
        hidden_states, n = self.norm1.skip(hidden_states, defer=d1)
        hidden_states = self.attn1(n, encoder_hidden_states=None, attention_mask=attention_mask, _residual=hidden_states, **kw)
        if cfg_expand:          # shared classifier-free-guidance prefix ends here: the text cross-attention is the first op that tells the halves apart
            hidden_states = torch.cat([hidden_states, hidden_states], dim=0)
        if fuse2:
            hidden_states = self._xattn_fused(hidden_states, encoder_hidden_states, kw)
        elif self.attn2 is not None:
            hidden_states, n = self.norm2.skip(hidden_states, defer=d2)
            hidden_states = self.attn2(n, encoder_hidden_states=encoder_hidden_states, attention_mask=encoder_attention_mask,
                                       _residual=hidden_states, **kw)
        if not torch.is_grad_enabled():
            y = self.ff.forward_ln(hidden_states, self.norm3, hidden_states, tail=tail)
            if y is not None:
                return y
        hidden_states, n = self.norm3.skip(hidden_states, defer=True)
        return self.ff(n, residual=hidden_states)


class Transformer2DModelOutput:
    def __init__(self, sample):
        self.sample = sample


class Transformer2DModel(nn.Module):
    """GN(eps 1e-6) -> 1x1 conv -> tokens -> BasicTransformerBlock -> 1x1 conv -> + residual
    (ctor args at fmc/models/unet_blocks.py:323-333)."""

    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=None, out_channels=None,
                 num_layers=1, dropout=0.0, norm_num_groups=32, cross_attention_dim=None, attention_bias=False,
                 activation_fn="geglu", use_linear_projection=False, only_cross_attention=False,
                 double_self_attention=False, upcast_attention=False, norm_type="layer_norm",
                 norm_elementwise_affine=True):
        super().__init__()
        assert not use_linear_projection, "SD-1.5 / FMC use conv projections"
        inner = num_attention_heads * attention_head_dim
        self.in_channels = in_channels
        self.norm = GroupNorm(num_groups=norm_num_groups, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = Conv2d(in_channels, inner, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, dropout=dropout,
                                  cross_attention_dim=cross_attention_dim, activation_fn=activation_fn,
                                  attention_bias=attention_bias, upcast_attention=upcast_attention)
            for _ in range(num_layers)])
        self.proj_out = Conv2d(inner, in_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, class_labels=None,
                cross_attention_kwargs=None, attention_mask=None, encoder_attention_mask=None,
                return_dict: bool = True, cfg_expand: bool = False):
        """`cfg_expand`: `hidden_states` holds ONE copy of the two identical halves of a classifier-free-guidance batch; everything
        up to and including the self-attention runs once, the output covers both halves (see the `cfg_shared_input` keyword of UNet3DConditionModel.forward)."""
        n, c, h, w = hidden_states.shape
        residual = to_tokens(hidden_states)
        if NORM_SKIP and torch.is_grad_enabled() and residual.requires_grad and residual.is_cuda:
            # the norm and the `+ residual` of proj_out as one autograd node (hip_ops.groupnorm_silu_skip)
            residual, x = K.groupnorm_silu_skip(residual if residual.is_contiguous() else residual.contiguous(), f32_param(self.norm, "weight"),
                                                f32_param(self.norm, "bias"), self.norm.num_groups, self.norm.eps, False)
        else:
            x = None
        w_in = self.proj_in.weight.view(self.proj_in.out_channels, c)
        ln_in = None if torch.is_grad_enabled() else self.transformer_blocks[0].norm1.ln_spec(
            stats_only=type(self.transformer_blocks[0].attn1.processor).__name__ in ("AttnProcessor", "LoRAAttnProcessor"))
        tag = getattr(hidden_states, "_fmc_gn", None)
        if x is None and K.gn_fold_ok(residual, tag, self.norm.num_groups, w_in, ln_in):
            # the norm folded into per-image weights of proj_in: the normalised tensor is neither written nor read (hip_ops.linear_gnfold)
            x = K.linear_gnfold(residual, tag, f32_param(self.norm, "weight"), f32_param(self.norm, "bias"), self.norm.num_groups, self.norm.eps,
                                w_in, self.proj_in.bias, ln_in)
        else:
            if x is None:
                x = K.groupnorm_silu(residual, f32_param(self.norm, "weight"), f32_param(self.norm, "bias"),
                                     self.norm.num_groups, self.norm.eps, False, gn_tag=tag)
            x = linear_op(x, w_in, self.proj_in.bias, ln=ln_in)
        if cfg_expand:
            residual = torch.cat([residual, residual], dim=0)
        wp = self.proj_out.weight.view(c, self.proj_out.in_channels)
        last = len(self.transformer_blocks) - 1
        for bi, blk in enumerate(self.transformer_blocks):
            x = blk(x, attention_mask=attention_mask, encoder_hidden_states=encoder_hidden_states,
                    encoder_attention_mask=encoder_attention_mask, timestep=timestep,
                    cross_attention_kwargs=cross_attention_kwargs, class_labels=class_labels,
                    **({"cfg_expand": True} if (cfg_expand and bi == 0) else {}),
                    # (the last block's feed-forward may take proj_out + residual into its own last launch: hip_ops.ff_tail)
                    **({"tail": (wp, self.proj_out.bias, residual, h * w)} if (bi == last and not torch.is_grad_enabled()) else {}))
        if not getattr(x, "_fmc_tail", False):
            x = linear_op(x, wp, self.proj_out.bias, residual, gn_hw=h * w)      # (the motion module / next ResNet block opens with a GroupNorm)
        out = from_tokens(x, h, w)
        return Transformer2DModelOutput(out) if return_dict else (out,)
