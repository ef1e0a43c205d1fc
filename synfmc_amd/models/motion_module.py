"""Temporal (motion) modules of FMC on the gfx950 kernels.

Mirrors `fmc/models/motion_module.py` (class names, constructor arguments, parameter names), with the
layout change that removes the reference's transposes: the reference moves `b c f h w` to `(b h w) f c`
and back around every module (:218, :232); here the video stays channels-last `[B, F, (h w), C]`, every
per-token op (LayerNorm, projections, GEGLU FF) runs on that buffer as is, and `fmc_temporal_attn_fwd`
walks the frame axis with a stride.  The reference's 3-D `(b h w) f c` token layout is accepted as well
(the camera encoder's public seam, `fmc/models/pose_adaptor.py:236-238`).
"""
from __future__ import annotations

import math
import os
from typing import Any, Dict, Optional

import torch
import torch.nn.functional as F
from torch import nn

from .. import hip_ops as K
from .attention_processor import (AttnProcessor, LORAPoseAdaptorAttnProcessor, LoRAAttnProcessor, PoseAdaptorAttnProcessor,
                                  _pose_tokens)
from .layers import Attention, FeedForward, LayerNorm, f32_param, linear_op
from .resnet import InflatedGroupNorm


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


class TemporalTransformer3DModelOutput:
    def __init__(self, sample):
        self.sample = sample


def enable_fp8_temporal_attention(module: nn.Module, enabled: bool = True, margin: float = 1.25) -> int:
    """Switch every `TemporalSelfAttention` under `module` (U-Net motion modules, camera encoder) to the fp8 path
    (BASELINE.json configs[4]): e4m3 q | k | v from the QKV projection's epilogue with per-tensor delayed scaling, QK^T on the
    fp8 MFMA (`fmc_linear_fp8_qkv`, `fmc_temporal_attn_fp8_fwd/_bwd`).  Returns the number of attention modules switched.
    bf16 activations only; modules whose width is not a multiple of 64 keep the bf16 kernel."""
    n = 0
    for m in module.modules():
        if isinstance(m, TemporalSelfAttention):
            if enabled:
                dev = next(m.parameters()).device
                m.__dict__["_fp8_scales"] = K.Fp8QKVScales(dev, margin)
            else:
                m.__dict__.pop("_fp8_scales", None)
            n += 1
    return n


def get_motion_module(in_channels, motion_module_type: str, motion_module_kwargs: dict):
    if motion_module_type == "Vanilla":
        return VanillaTemporalModule(in_channels=in_channels, **motion_module_kwargs)
    raise ValueError


TEMPORAL_FUSED = os.environ.get("FMC_TEMPORAL_FUSED", "1") != "0"     # A/B switch: the fused temporal attention block kernel (40x64 level)


class PositionalEncoding(nn.Module):
    """Sinusoidal table, buffer `pe` `[1, max_len, C]` (motion_module.py:303-321)."""

    def __init__(self, d_model, dropout=0.0, max_len=32):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(1, max_len, d_model)
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)

    def table(self) -> torch.Tensor:
        """fp32 `[max_len, C]` table for the fused LayerNorm+PE kernel."""
        pe = self.pe
        if pe.dtype != torch.float32:
            hit = self.__dict__.get("_pe32")
            if hit is None or hit[0] != (pe.data_ptr(), pe._version):
                hit = ((pe.data_ptr(), pe._version), pe.float())
                self.__dict__["_pe32"] = hit
            pe = hit[1]
        return pe[0]

    def forward(self, x):
        if x.ndim == 4:                      # [B, F, P, C]
            return x + self.pe[0, : x.size(1)].to(x.dtype)[None, :, None, :]
        return x + self.pe[:, : x.size(1)].to(x.dtype)


class TemporalSelfAttention(Attention):
    """motion_module.py:324-389.  `forward` adds the positional encoding (unless the caller already fused it
    into the LayerNorm, `_pe_applied=True`), reshapes `pose_feature` and dispatches to the processor with
    `encoder_hidden_states=None`."""

    def __init__(self, attention_mode=None, temporal_position_encoding=False,
                 temporal_position_encoding_max_len=32, rescale_output_factor=1.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert attention_mode == "Temporal_Self"
        self.pos_encoder = PositionalEncoding(kwargs["query_dim"], max_len=temporal_position_encoding_max_len) \
            if temporal_position_encoding else None
        self.rescale_output_factor = rescale_output_factor

    def set_use_memory_efficient_attention_xformers(self, *a, **k):
        pass

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, _pe_applied: bool = False,
                **cross_attention_kwargs):
        if self.pos_encoder is not None and not _pe_applied:
            hidden_states = self.pos_encoder(hidden_states)
        kw = dict(cross_attention_kwargs)
        if isinstance(self.processor, PoseAdaptorAttnProcessor):
            pose_feature = kw.pop("pose_feature")
            return self.processor(self, hidden_states, pose_feature, encoder_hidden_states=None,
                                  attention_mask=attention_mask, temporal=True, **kw)
        return self.processor(self, hidden_states, encoder_hidden_states=None, attention_mask=attention_mask,
                              temporal=True, **kw)


class TemporalTransformerBlock(nn.Module):
    """motion_module.py:237-300: `x = attn_i(LN_i(x)) + x` for every attention block, then `x = FF(LN(x)) + x`.
    Tokens: native `[B, F, P, C]` or reference `[N, F, C]`."""

    def __init__(self, dim, num_attention_heads, attention_head_dim,
                 attention_block_types=("Temporal_Self", "Temporal_Self"), dropout=0.0, norm_num_groups=32,
                 cross_attention_dim=768, activation_fn="geglu", attention_bias=False, upcast_attention=False,
                 temporal_position_encoding=False, temporal_position_encoding_max_len=32,
                 encoder_hidden_states_query=(False, False), attention_activation_scale=1.0,
                 attention_processor_kwargs: Dict = {}, rescale_output_factor=1.0):
        super().__init__()
        self.attention_block_types = attention_block_types
        self.attention_blocks = nn.ModuleList([
            TemporalSelfAttention(
                attention_mode=name,
                cross_attention_dim=cross_attention_dim if name in ("Temporal_Cross", "Temporal_Pose_Adaptor") else None,
                query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim, dropout=dropout,
                bias=attention_bias, upcast_attention=upcast_attention,
                temporal_position_encoding=temporal_position_encoding,
                temporal_position_encoding_max_len=temporal_position_encoding_max_len,
                rescale_output_factor=rescale_output_factor)
            for name in attention_block_types])
        self.norms = nn.ModuleList([LayerNorm(dim) for _ in attention_block_types])
        self.ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn)
        self.ff_norm = LayerNorm(dim)

    # ---- the fused attention block (`fmc_temporal_block_bf16`, round 4) -------------------------------------------------------------------
    def fused_blocks_ok(self, hidden_states, attention_mask, cross_attention_kwargs) -> bool:
        """Every attention block of this transformer block can run as ONE launch each (LayerNorm + pe -> [Camera-Adapter merge] -> q | k | v ->
        attention over the frames -> out-projection + residual): inference, bf16 `[B, 16, P, 320]` tokens with P % 10 == 0 or `[B, 16, P, 640]` with P % 5 == 0, 8 heads, plain /
        Camera-Adapter / frozen-LoRA processors, nothing applied after the output projection.  Anything else keeps the un-fused chain."""
        if not TEMPORAL_FUSED or attention_mask is not None or hidden_states.ndim != 4:
            return False
        for blk in self.attention_blocks:
            proc = blk.processor
            if not isinstance(proc, (AttnProcessor, LoRAAttnProcessor, PoseAdaptorAttnProcessor, LORAPoseAdaptorAttnProcessor)):
                return False
            if not K.temporal_block_supported(hidden_states, blk.heads) or blk.inner_dim != hidden_states.shape[-1]:
                return False
            if (blk.residual_connection or blk.rescale_output_factor != 1.0 or blk.__dict__.get("_fp8_scales") is not None
                    or blk.to_q.weight.dtype != torch.bfloat16 or getattr(blk, "is_cross", False)):
                return False
            if isinstance(proc, (PoseAdaptorAttnProcessor, LORAPoseAdaptorAttnProcessor)):
                pf = cross_attention_kwargs.get("pose_feature")
                if pf is None or not (proc.query_condition and proc.key_value_condition) or proc.qkv_merge.weight.dtype != torch.bfloat16:
                    return False
                # (what the un-fused `_merge` falls back on -- a pose feature of another dtype or shape -- must fall back here too)
                if pf.dtype != torch.bfloat16 or not pf.is_cuda:
                    return False
                if pf.ndim == 5:
                    b, c, f, hh, ww = pf.shape
                    if (b, f, hh * ww, c) != tuple(hidden_states.shape):
                        return False
                elif tuple(pf.shape) != tuple(hidden_states.shape):
                    return False
        return True

    def _fused_constants(self, i, frames):
        """(gamma fp32, beta + pe rows fp32 `[frames, C]`) of attention block i, cached on the block."""
        norm, enc = self.norms[i], self.attention_blocks[i].pos_encoder
        g, b = f32_param(norm, "weight"), f32_param(norm, "bias")
        pe = None if enc is None else enc.table()
        key = (g.data_ptr(), g._version, b.data_ptr(), b._version, None if pe is None else (pe.data_ptr(), pe._version), frames)
        hit = self.__dict__.setdefault("_fused_c", {}).get(i)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                bpe = b[None, :].expand(frames, -1) if pe is None else b[None, :] + pe[:frames]
                hit = (key, g.contiguous(), bpe.float().contiguous())
            self.__dict__["_fused_c"][i] = hit
        return hit[1], hit[2]

    def _fused_weights(self, i, lora, lora_scale):
        attn = self.attention_blocks[i]
        w_qkv, _, w_o = attn.fused_weights(lora, lora_scale)
        key = (w_qkv.data_ptr(), w_qkv._version, w_o.data_ptr(), w_o._version)
        hit = attn.__dict__.get("_fused_tb")
        if hit is None or hit[0] != key:
            # (the entry keeps w_qkv / w_o alive: an equal data pointer then means the same storage, not an address the caching allocator
            #  handed to the NEXT processor's merged weights; `Attention.set_processor` drops the entry as well)
            if w_o.shape[0] == 640:                      # the 20x32 level: weights in MFMA-fragment order (temporal_block640.hip)
                hit = (key, K.pack_temporal_qkv80(w_qkv, attn.heads), K.pack_w_frag80(w_o), (w_qkv, w_o))
            else:
                hit = (key, K.pack_temporal_qkv(w_qkv, attn.heads), K._w_tilemajor(w_o), (w_qkv, w_o))
            attn.__dict__["_fused_tb"] = hit
        return hit[1], hit[2]

    @staticmethod
    def _merge_packed(proc, wm):
        if wm.shape[0] != 640:
            return K._w_tilemajor(wm)
        key = (wm.data_ptr(), wm._version)
        hit = proc.__dict__.get("_fused_wm")
        if hit is None or hit[0] != key:
            hit = (key, K.pack_w_frag80(wm), wm)
            proc.__dict__["_fused_wm"] = hit
        return hit[1]

    def _forward_fused(self, hidden_states, cross_attention_kwargs, tail=None):
        frames = hidden_states.shape[1]
        h = hidden_states if hidden_states.is_contiguous() else hidden_states.contiguous()
        n_blocks = len(self.attention_blocks)
        stats = None
        for i, attn in enumerate(self.attention_blocks):
            proc = attn.processor
            lora = proc if isinstance(proc, (LoRAAttnProcessor, LORAPoseAdaptorAttnProcessor)) else None
            lora_scale = 1.0
            if lora is not None:
                from .attention_processor import _require_frozen, resolve_lora_scale
                _require_frozen(proc)
                # the same rule as the un-fused processors' `scale` defaults (a missing kwarg is NOT `lora_scale` for the LoRA + pose processor)
                lora_scale = (resolve_lora_scale(proc, cross_attention_kwargs["scale"]) if "scale" in cross_attention_kwargs
                              else resolve_lora_scale(proc))
            gamma, bpe = self._fused_constants(i, frames)
            w_qkv, w_o = self._fused_weights(i, lora, lora_scale)
            kw = {}
            if isinstance(proc, (PoseAdaptorAttnProcessor, LORAPoseAdaptorAttnProcessor)):
                # `merge(h + pose) * s + h` (attention_processor.py:256-258): the pose part s (W pose + b) once per clip, the rest in the kernel
                s = proc.scale if isinstance(proc, LORAPoseAdaptorAttnProcessor) else (cross_attention_kwargs.get("scale") or proc.scale)
                pose = _pose_tokens(cross_attention_kwargs["pose_feature"], h)
                assert pose.shape == h.shape, "pose_feature does not match the hidden states"
                wm, bm = proc.qkv_merge.weight, proc.qkv_merge.bias
                kw = dict(w_merge_tm=self._merge_packed(proc, wm), pose_term=proc._pose_term(pose, wm, bm, s), merge_scale=s)
            # (row statistics for the feed-forward's LayerNorm-in-GEMM: the 40x64 level, and only while the feed-forward does not normalise its input itself)
            last = i == n_blocks - 1 and h.shape[-1] == 320 and not K.geglu_ln_direct_ok(h, self.ff.net[0].proj.weight)
            out = K.temporal_block(h, gamma, bpe, self.norms[i].eps, w_qkv, w_o, attn.to_out[0].bias, attn.scale,
                                   stats_eps=self.ff_norm.eps if last else None, **kw)
            h, stats = out if last else (out, None)
        if stats is not None:
            # the feed-forward's norm is applied by its GEGLU projection from the row statistics the last block left
            K.ln_epilogue_calls["emitted"] += 1
            h._fmc_ln = (stats, self.ff_norm._ln_key(None, 1, 1), True)
        if stats is None:
            y = self.ff.forward_ln(h, self.ff_norm, h, tail=tail)       # (LayerNorm + GEGLU projection as one launch; `tail`: + proj_out, hip_ops.ff_tail)
            if y is not None:
                return y
        hidden_states, n = self.ff_norm.skip(h, defer=True)
        return self.ff(n, residual=hidden_states)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None,
                cross_attention_kwargs: Dict[str, Any] = {}, tail=None):
        if not torch.is_grad_enabled() and self.fused_blocks_ok(hidden_states, attention_mask, cross_attention_kwargs):
            return self._forward_fused(hidden_states, cross_attention_kwargs, tail)
        if hidden_states.ndim == 4:
            frames, inner = hidden_states.shape[1], hidden_states.shape[2]
        else:
            frames, inner = hidden_states.shape[1], 1
        def spec(i):                                    # the norm (+ positional encoding) in front of attention block i, or the FF norm
            if torch.is_grad_enabled():
                return None
            if i >= len(self.attention_blocks):
                return self.ff_norm.ln_spec(stats_only=True)         # (feeds the GEGLU projection only: applied in that GEMM's epilogue)
            enc = self.attention_blocks[i].pos_encoder
            return self.norms[i].ln_spec() if enc is None else self.norms[i].ln_spec(enc.table(), inner, frames)
        for i, attention_block in enumerate(self.attention_blocks):
            attention_block.__dict__["_next_ln"] = spec(i + 1)
            attention_block.__dict__["_lazy_res"] = not torch.is_grad_enabled()      # (the next consumer is `norm.skip` / `ff_norm.skip` below)
        for attention_block, norm in zip(self.attention_blocks, self.norms):
            pe = attention_block.pos_encoder
            if pe is not None:       # LayerNorm and `pos_encoder(norm(x))` in one pass (motion_module.py:288,355)
                hidden_states, n = norm.skip(hidden_states, pe=pe.table(), pe_inner=inner, pe_frames=frames)
            else:
                hidden_states, n = norm.skip(hidden_states)
            hidden_states = attention_block(n, encoder_hidden_states=None, attention_mask=attention_mask,
                                            _pe_applied=True, _residual=hidden_states, **cross_attention_kwargs)
        hidden_states, n = self.ff_norm.skip(hidden_states, defer=True)
        return self.ff(n, residual=hidden_states)


class TemporalTransformer3DModel(nn.Module):
    """motion_module.py:93-234."""

    def __init__(self, in_channels, num_attention_heads, attention_head_dim, num_layers,
                 attention_block_types=("Temporal_Self", "Temporal_Self"), dropout=0.0, norm_num_groups=32,
                 cross_attention_dim=320, activation_fn="geglu", attention_bias=False, upcast_attention=False,
                 temporal_position_encoding=False, temporal_position_encoding_max_len=32,
                 encoder_hidden_states_query=(False, False), attention_activation_scale=1.0,
                 attention_processor_kwargs: Dict = {}, causal_temporal_attention=None,
                 causal_temporal_attention_mask_type="", rescale_output_factor=1.0):
        super().__init__()
        assert causal_temporal_attention is not None
        if causal_temporal_attention:
            raise NotImplementedError("causal temporal masks (motion_module.py:151-208) are unused by the FMC configs")
        self.causal_temporal_attention = causal_temporal_attention
        inner_dim = num_attention_heads * attention_head_dim
        self.norm = InflatedGroupNorm(num_groups=norm_num_groups, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner_dim)
        self.transformer_blocks = nn.ModuleList([
            TemporalTransformerBlock(
                dim=inner_dim, num_attention_heads=num_attention_heads, attention_head_dim=attention_head_dim,
                attention_block_types=attention_block_types, dropout=dropout, norm_num_groups=norm_num_groups,
                cross_attention_dim=cross_attention_dim, activation_fn=activation_fn, attention_bias=attention_bias,
                upcast_attention=upcast_attention, temporal_position_encoding=temporal_position_encoding,
                temporal_position_encoding_max_len=temporal_position_encoding_max_len,
                rescale_output_factor=rescale_output_factor)
            for _ in range(num_layers)])
        self.proj_out = nn.Linear(inner_dim, in_channels)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None,
                cross_attention_kwargs: Dict[str, Any] = {}):
        assert hidden_states.dim() == 5, f"Expected hidden_states to have ndim=5, but got ndim={hidden_states.dim()}."
        b, c, f, h, w = hidden_states.shape
        t = hidden_states.permute(0, 2, 3, 4, 1)                 # [B, F, h, w, C]; free on channels_last_3d storage
        if not t.is_contiguous():
            t = t.contiguous()
        residual = t.view(b * f, h * w, c)
        from .layers import NORM_SKIP
        if NORM_SKIP and torch.is_grad_enabled() and residual.requires_grad and residual.is_cuda:
            # the norm and the `+ residual` of proj_out as one autograd node (hip_ops.groupnorm_silu_skip)
            residual, x = K.groupnorm_silu_skip(residual, f32_param(self.norm, "weight"), f32_param(self.norm, "bias"),
                                                self.norm.num_groups, self.norm.eps, False)
        else:
            x = None
        blk0 = self.transformer_blocks[0]
        ln0 = None
        # (the fused attention blocks normalise their input themselves: proj_in then has no LayerNorm to emit; the residual stands in for its output here --
        #  same shape / dtype / device when the inner width equals the channel count, the only case the fused block takes)
        will_fuse = (not torch.is_grad_enabled() and c == self.proj_in.out_features and residual.is_contiguous()
                     and blk0.fused_blocks_ok(residual.view(b, f, h * w, c), attention_mask, cross_attention_kwargs))
        if not torch.is_grad_enabled() and not will_fuse:   # the first block's first norm (+ PE) leaves proj_in's epilogue
            enc0 = blk0.attention_blocks[0].pos_encoder
            ln0 = blk0.norms[0].ln_spec() if enc0 is None else blk0.norms[0].ln_spec(enc0.table(), h * w, f)
        tag = getattr(hidden_states, "_fmc_gn", None)
        if x is None and K.gn_fold_ok(residual, tag, self.norm.num_groups, self.proj_in.weight, ln0):
            # the norm folded into per-image weights of proj_in: the normalised tensor is neither written nor read (hip_ops.linear_gnfold)
            x = K.linear_gnfold(residual, tag, f32_param(self.norm, "weight"), f32_param(self.norm, "bias"), self.norm.num_groups, self.norm.eps,
                                self.proj_in.weight, self.proj_in.bias, ln0)
        else:
            if x is None:
                x = K.groupnorm_silu(residual, f32_param(self.norm, "weight"), f32_param(self.norm, "bias"),
                                     self.norm.num_groups, self.norm.eps, False, gn_tag=tag)
            x = linear_op(x, self.proj_in.weight, self.proj_in.bias, ln=ln0)
        x = K.carry_ln(x, x.view(b, f, h * w, -1))
        last = len(self.transformer_blocks) - 1
        for bi, block in enumerate(self.transformer_blocks):
            x = block(x, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask,
                      cross_attention_kwargs=cross_attention_kwargs,
                      # (the last block's feed-forward may take proj_out + residual into its own last launch: hip_ops.ff_tail)
                      **({"tail": (self.proj_out.weight, self.proj_out.bias, residual, h * w)} if (bi == last and not torch.is_grad_enabled()) else {}))
        if not getattr(x, "_fmc_tail", False):
            x = linear_op(x.view(b * f, h * w, -1), self.proj_out.weight, self.proj_out.bias, residual, gn_hw=h * w)
        return K.carry_gn(x, x.view(b, f, h, w, c).permute(0, 4, 1, 2, 3))   # (the next ResNet block / conv_norm_out opens with a GroupNorm)


class VanillaTemporalModule(nn.Module):
    """motion_module.py:44-90."""

    def __init__(self, in_channels, num_attention_heads=8, num_transformer_block=2,
                 attention_block_types=("Temporal_Self",), temporal_position_encoding=True,
                 temporal_position_encoding_max_len=32, temporal_attention_dim_div=1, cross_attention_dim=320,
                 zero_initialize=True, encoder_hidden_states_query=(False, False), attention_activation_scale=1.0,
                 attention_processor_kwargs: Dict = {}, causal_temporal_attention=False,
                 causal_temporal_attention_mask_type="", rescale_output_factor=1.0):
        super().__init__()
        self.temporal_transformer = TemporalTransformer3DModel(
            in_channels=in_channels, num_attention_heads=num_attention_heads,
            attention_head_dim=in_channels // num_attention_heads // temporal_attention_dim_div,
            num_layers=num_transformer_block, attention_block_types=tuple(attention_block_types),
            cross_attention_dim=cross_attention_dim, temporal_position_encoding=temporal_position_encoding,
            temporal_position_encoding_max_len=temporal_position_encoding_max_len,
            encoder_hidden_states_query=encoder_hidden_states_query,
            attention_activation_scale=attention_activation_scale,
            attention_processor_kwargs=attention_processor_kwargs,
            causal_temporal_attention=causal_temporal_attention,
            causal_temporal_attention_mask_type=causal_temporal_attention_mask_type,
            rescale_output_factor=rescale_output_factor)
        if zero_initialize:
            self.temporal_transformer.proj_out = zero_module(self.temporal_transformer.proj_out)

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None,
                cross_attention_kwargs: Dict[str, Any] = {}):
        return self.temporal_transformer(hidden_states, encoder_hidden_states, attention_mask,
                                         cross_attention_kwargs=cross_attention_kwargs)
