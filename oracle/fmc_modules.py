"""CPU restatement of the reference-owned modules on the FMC denoising hot path.

TEST INFRASTRUCTURE (see `oracle/__init__.py`).  fp32, plain PyTorch, reference
tensor layout (`b c f h w` outside, `(b f) c h w` / `(b h w) f c` inside) so
every intermediate can be compared with the reference one-to-one.  Module and
parameter names equal the reference's, so `state_dict()` keys are identical
(SURVEY.md Appendix B); bodies are a restatement, each citing what it follows.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F
from einops import rearrange, repeat
from torch import nn

from . import diffusers_restated as D


# ----------------------------------------------------------------------------
# inflated 2-D layers  (fmc/models/resnet.py:16-37)
# ----------------------------------------------------------------------------
class InflatedConv3d(nn.Conv2d):
    """Per-frame 2-D conv on a `b c f h w` video (resnet.py:16-24)."""

    def forward(self, x):
        f = x.shape[2]
        y = super().forward(rearrange(x, "b c f h w -> (b f) c h w"))
        return rearrange(y, "(b f) c h w -> b c f h w", f=f)


class InflatedGroupNorm(nn.GroupNorm):
    """Per-frame GroupNorm (resnet.py:27-37): statistics never mix frames."""

    def forward(self, x):
        f = x.shape[2]
        y = super().forward(rearrange(x, "b c f h w -> (b f) c h w"))
        return rearrange(y, "(b f) c h w -> b c f h w", f=f)


# ----------------------------------------------------------------------------
# attention processors  (fmc/models/attention_processor.py)
# ----------------------------------------------------------------------------
def _attend(attn, q_in, kv_in, attention_mask, proj_q, proj_k, proj_v, proj_o):
    """Shared tail of all four processors: project, split heads, softmax(QK^T*scale)V,
    merge heads, out-proj (+dropout p=0), optional residual handled by caller."""
    q = attn.head_to_batch_dim(proj_q(q_in))
    k = attn.head_to_batch_dim(proj_k(kv_in))
    v = attn.head_to_batch_dim(proj_v(kv_in))
    probs = attn.get_attention_scores(q, k, attention_mask)
    out = attn.batch_to_head_dim(torch.bmm(probs, v))
    return attn.to_out[1](proj_o(out))


def _tokens(x):
    """`b c h w -> b (h w) c` for 4-D inputs, 3-D passes through.  The reference's
    5-D branch tests `hidden_states.dim == 5` (a bound method vs an int: never
    true, attention_processor.py:220) so 5-D inputs fall into the assert."""
    if x.ndim == 4:
        return rearrange(x, "b c h w -> b (h w) c")
    assert x.ndim == 3
    return x


class AttnProcessor:
    """attention_processor.py:15-82.  `pose_feature` is accepted and ignored (:28)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 scale: float = 1.0, pose_feature=None):
        residual = hidden_states
        shape4 = hidden_states.shape if hidden_states.ndim == 4 else None
        if shape4 is not None:
            hidden_states = hidden_states.flatten(2).transpose(1, 2)
        b, s_kv, _ = (hidden_states if encoder_hidden_states is None else encoder_hidden_states).shape
        attention_mask = attn.prepare_attention_mask(attention_mask, s_kv, b)
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        out = _attend(attn, hidden_states, ctx, attention_mask,
                      lambda x: attn.to_q(x, scale), lambda x: attn.to_k(x, scale),
                      lambda x: attn.to_v(x, scale), lambda x: attn.to_out[0](x, scale))
        if shape4 is not None:
            out = out.transpose(-1, -2).reshape(shape4)
        if attn.residual_connection:
            out = out + residual
        return out / attn.rescale_output_factor


class LoRAAttnProcessor(nn.Module):
    """attention_processor.py:85-169: every projection is `W x + s * up(down(x))`."""

    def __init__(self, hidden_size=None, cross_attention_dim=None, rank=4, network_alpha=None, lora_scale=1.0):
        super().__init__()
        self.rank, self.lora_scale = rank, lora_scale
        kv_in = cross_attention_dim or hidden_size
        self.to_q_lora = D.LoRALinearLayer(hidden_size, hidden_size, rank, network_alpha)
        self.to_k_lora = D.LoRALinearLayer(kv_in, hidden_size, rank, network_alpha)
        self.to_v_lora = D.LoRALinearLayer(kv_in, hidden_size, rank, network_alpha)
        self.to_out_lora = D.LoRALinearLayer(hidden_size, hidden_size, rank, network_alpha)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 pose_feature=None, scale=None):
        s = self.lora_scale if scale is None else scale
        residual = hidden_states
        shape4 = hidden_states.shape if hidden_states.ndim == 4 else None
        if shape4 is not None:
            hidden_states = hidden_states.flatten(2).transpose(1, 2)
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        attention_mask = attn.prepare_attention_mask(attention_mask, ctx.shape[1], ctx.shape[0])
        out = _attend(attn, hidden_states, ctx, attention_mask,
                      lambda x: attn.to_q(x) + s * self.to_q_lora(x),
                      lambda x: attn.to_k(x) + s * self.to_k_lora(x),
                      lambda x: attn.to_v(x) + s * self.to_v_lora(x),
                      lambda x: attn.to_out[0](x) + s * self.to_out_lora(x))
        if shape4 is not None:
            out = out.transpose(-1, -2).reshape(shape4)
        if attn.residual_connection:
            out = out + residual
        return out / attn.rescale_output_factor


class _PoseMergeMixin:
    """Zero-initialised merge layers of the Camera Adapter (attention_processor.py:187-200)."""

    def _build_merge(self, hidden_size, pose_feature_dim, query_condition, key_value_condition):
        assert hidden_size == pose_feature_dim
        self.query_condition, self.key_value_condition = query_condition, key_value_condition
        name = "qkv_merge" if (query_condition and key_value_condition) else ("q_merge" if query_condition else "kv_merge")
        layer = nn.Linear(hidden_size, hidden_size)
        nn.init.zeros_(layer.weight)
        nn.init.zeros_(layer.bias)
        setattr(self, name, layer)

    def _merge(self, hidden_states, encoder_hidden_states, pose_feature, s):
        """attention_processor.py:256-265: returns (query_in, key_value_in)."""
        if self.query_condition and self.key_value_condition:
            m = self.qkv_merge(hidden_states + pose_feature) * s + hidden_states
            return m, m
        if self.query_condition:
            return self.q_merge(hidden_states + pose_feature) * s + hidden_states, encoder_hidden_states
        kv = self.kv_merge(encoder_hidden_states + pose_feature) * s + encoder_hidden_states
        return hidden_states, kv


class PoseAdaptorAttnProcessor(nn.Module, _PoseMergeMixin):
    """Camera Adapter (attention_processor.py:172-293).  Note `forward` takes
    `pose_feature` positionally, right after `hidden_states` (:202-205)."""

    def __init__(self, hidden_size, pose_feature_dim=None, cross_attention_dim=None, query_condition=False,
                 key_value_condition=False, scale=1.0):
        super().__init__()
        self.hidden_size, self.pose_feature_dim = hidden_size, pose_feature_dim
        self.cross_attention_dim, self.scale = cross_attention_dim, scale
        self._build_merge(hidden_size, pose_feature_dim, query_condition, key_value_condition)

    def forward(self, attn, hidden_states, pose_feature, encoder_hidden_states=None, attention_mask=None,
                temb=None, scale=None):
        assert pose_feature is not None
        s = scale or self.scale                      # :211 (a LoRA "scale" kwarg would be reused here)
        residual = hidden_states
        hidden_states = _tokens(hidden_states)
        if self.query_condition and self.key_value_condition:
            assert encoder_hidden_states is None
        ctx = _tokens(hidden_states if encoder_hidden_states is None else encoder_hidden_states)
        pose_feature = _tokens(pose_feature)
        attention_mask = attn.prepare_attention_mask(attention_mask, ctx.shape[1], ctx.shape[0])
        q_in, kv_in = self._merge(hidden_states, ctx, pose_feature, s)
        out = _attend(attn, q_in, kv_in, attention_mask, attn.to_q, attn.to_k, attn.to_v, attn.to_out[0])
        if attn.residual_connection:
            out = out + residual
        return out / attn.rescale_output_factor


class LORAPoseAdaptorAttnProcessor(nn.Module, _PoseMergeMixin):
    """attention_processor.py:296-420: pose merge (always `self.scale`, :381-389) + LoRA projections."""

    def __init__(self, hidden_size, pose_feature_dim=None, cross_attention_dim=None, query_condition=False,
                 key_value_condition=False, scale=1.0, rank=4, network_alpha=None, lora_scale=1.0):
        super().__init__()
        self.hidden_size, self.pose_feature_dim = hidden_size, pose_feature_dim
        self.cross_attention_dim, self.scale = cross_attention_dim, scale
        self._build_merge(hidden_size, pose_feature_dim, query_condition, key_value_condition)
        self.rank, self.lora_scale = rank, lora_scale
        kv_in = cross_attention_dim or hidden_size
        self.to_q_lora = D.LoRALinearLayer(hidden_size, hidden_size, rank, network_alpha)
        self.to_k_lora = D.LoRALinearLayer(kv_in, hidden_size, rank, network_alpha)
        self.to_v_lora = D.LoRALinearLayer(kv_in, hidden_size, rank, network_alpha)
        self.to_out_lora = D.LoRALinearLayer(hidden_size, hidden_size, rank, network_alpha)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 scale=1.0, pose_feature=None):
        assert pose_feature is not None
        ls = self.lora_scale if scale is None else scale
        residual = hidden_states
        hidden_states = _tokens(hidden_states)
        if self.query_condition and self.key_value_condition:
            assert encoder_hidden_states is None
        ctx = _tokens(hidden_states if encoder_hidden_states is None else encoder_hidden_states)
        pose_feature = _tokens(pose_feature)
        attention_mask = attn.prepare_attention_mask(attention_mask, ctx.shape[1], ctx.shape[0])
        q_in, kv_in = self._merge(hidden_states, ctx, pose_feature, self.scale)
        out = _attend(attn, q_in, kv_in, attention_mask,
                      lambda x: attn.to_q(x) + ls * self.to_q_lora(x),
                      lambda x: attn.to_k(x) + ls * self.to_k_lora(x),
                      lambda x: attn.to_v(x) + ls * self.to_v_lora(x),
                      lambda x: attn.to_out[0](x) + ls * self.to_out_lora(x))
        if attn.residual_connection:
            out = out + residual
        return out / attn.rescale_output_factor


# ----------------------------------------------------------------------------
# motion module  (fmc/models/motion_module.py)
# ----------------------------------------------------------------------------
class PositionalEncoding(nn.Module):
    """Sinusoidal table `[1, max_len, C]`, even channels sin / odd cos (motion_module.py:303-321)."""

    def __init__(self, d_model, dropout=0.0, max_len=32):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pos = torch.arange(max_len).unsqueeze(1)
        div = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(1, max_len, d_model)
        pe[0, :, 0::2] = torch.sin(pos * div)
        pe[0, :, 1::2] = torch.cos(pos * div)
        self.register_buffer("pe", pe)

    def forward(self, x):
        return self.dropout(x + self.pe[:, : x.size(1)])


class TemporalSelfAttention(D.Attention):
    """motion_module.py:324-389.  PE is added to the (already LayerNorm-ed) input, so it
    reaches the merge layer, Q, K and V; `encoder_hidden_states` passed by the block is
    discarded (:370,378)."""

    def __init__(self, attention_mode=None, temporal_position_encoding=False,
                 temporal_position_encoding_max_len=32, rescale_output_factor=1.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert attention_mode == "Temporal_Self"
        self.pos_encoder = (PositionalEncoding(kwargs["query_dim"], max_len=temporal_position_encoding_max_len)
                            if temporal_position_encoding else None)
        self.rescale_output_factor = rescale_output_factor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        if self.pos_encoder is not None:
            hidden_states = self.pos_encoder(hidden_states)
        kw = dict(cross_attention_kwargs)
        if "pose_feature" in kw:
            pf = kw["pose_feature"]
            if pf.ndim == 5:
                pf = rearrange(pf, "b c f h w -> (b h w) f c")
            else:
                assert pf.ndim == 3
            kw["pose_feature"] = pf
        if isinstance(self.processor, PoseAdaptorAttnProcessor):
            pf = kw.pop("pose_feature")
            return self.processor(self, hidden_states, pf, encoder_hidden_states=None,
                                  attention_mask=attention_mask, **kw)
        return self.processor(self, hidden_states, encoder_hidden_states=None,
                              attention_mask=attention_mask, **kw)


class TemporalTransformerBlock(nn.Module):
    """motion_module.py:237-300: for each attention block `x = attn(LN(x)) + x`; then
    `x = FF(LN(x)) + x`.  Both blocks get the same kwargs (:289-295)."""

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
        self.norms = nn.ModuleList([nn.LayerNorm(dim) for _ in attention_block_types])
        self.ff = D.FeedForward(dim, dropout=dropout, activation_fn=activation_fn)
        self.ff_norm = nn.LayerNorm(dim)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None,
                cross_attention_kwargs: Dict = {}):
        for blk, norm, kind in zip(self.attention_blocks, self.norms, self.attention_block_types):
            n = norm(hidden_states)
            hidden_states = blk(n, encoder_hidden_states=n if kind == "Temporal_Self" else encoder_hidden_states,
                                attention_mask=attention_mask, **cross_attention_kwargs) + hidden_states
        return self.ff(self.ff_norm(hidden_states)) + hidden_states


class TemporalTransformer3DModel(nn.Module):
    """motion_module.py:93-234: per-frame GN(eps 1e-6) -> `(b h w) f c` -> Linear -> blocks ->
    Linear -> back -> + residual.  Causal masks (:151-208) are never enabled by the shipped
    configs (`causal_temporal_attention=False`) and are not restated."""

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
            raise NotImplementedError("causal temporal masks are unused by the shipped FMC configs")
        inner = num_attention_heads * attention_head_dim
        self.norm = InflatedGroupNorm(num_groups=norm_num_groups, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([
            TemporalTransformerBlock(
                dim=inner, num_attention_heads=num_attention_heads, attention_head_dim=attention_head_dim,
                attention_block_types=attention_block_types, dropout=dropout, norm_num_groups=norm_num_groups,
                cross_attention_dim=cross_attention_dim, activation_fn=activation_fn,
                attention_bias=attention_bias, upcast_attention=upcast_attention,
                temporal_position_encoding=temporal_position_encoding,
                temporal_position_encoding_max_len=temporal_position_encoding_max_len,
                rescale_output_factor=rescale_output_factor)
            for _ in range(num_layers)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None,
                cross_attention_kwargs: Dict = {}):
        assert hidden_states.dim() == 5
        residual = hidden_states
        h, w = hidden_states.shape[-2:]
        x = rearrange(self.norm(hidden_states), "b c f h w -> (b h w) f c")
        x = self.proj_in(x)
        for blk in self.transformer_blocks:
            x = blk(x, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask,
                    cross_attention_kwargs=cross_attention_kwargs)
        x = self.proj_out(x)
        return rearrange(x, "(b h w) f c -> b c f h w", h=h, w=w) + residual


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
            causal_temporal_attention=causal_temporal_attention,
            causal_temporal_attention_mask_type=causal_temporal_attention_mask_type,
            rescale_output_factor=rescale_output_factor)
        if zero_initialize:
            for p in self.temporal_transformer.proj_out.parameters():
                p.detach().zero_()

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None,
                cross_attention_kwargs: Dict = {}):
        return self.temporal_transformer(hidden_states, encoder_hidden_states, attention_mask,
                                         cross_attention_kwargs=cross_attention_kwargs)


def get_motion_module(in_channels, motion_module_type, motion_module_kwargs):
    if motion_module_type != "Vanilla":
        raise ValueError
    return VanillaTemporalModule(in_channels=in_channels, **motion_module_kwargs)


# ----------------------------------------------------------------------------
# U-Net blocks  (fmc/models/unet_blocks.py, fmc/modified_modules.py)
# ----------------------------------------------------------------------------
def _per_frame(fn, x, *args):
    f = x.shape[2]
    y = fn(rearrange(x, "b c f h w -> (b f) c h w"), *args)
    return rearrange(y, "(b f) c h w -> b c f h w", f=f)


def _resnet(cin, cout, temb, eps, groups, act, scale=1.0):
    return D.ResnetBlock2D(in_channels=cin, out_channels=cout, temb_channels=temb, eps=eps, groups=groups,
                           dropout=0.0, time_embedding_norm="default", non_linearity=act,
                           output_scale_factor=scale, pre_norm=True)


def _transformer(heads, channels, cross_dim, groups):
    return D.Transformer2DModel(heads, channels // heads, in_channels=channels, num_layers=1,
                                cross_attention_dim=cross_dim, norm_num_groups=groups,
                                use_linear_projection=False, only_cross_attention=False, upcast_attention=False)


class _Block3D(nn.Module):
    """One (resnet, [spatial transformer], [motion module]) layer stack shared by the five
    reference block types.  Layer order follows unet_blocks.py:397-412 / :520-527."""

    has_cross_attention = False

    def _layer(self, i, hidden_states, temb_rep, encoder_hidden_states, cross_kw, motion_kw):
        hidden_states = _per_frame(self.resnets[i], hidden_states, temb_rep)
        if self.has_cross_attention:
            attn = self.attentions[i]
            hidden_states = _per_frame(
                lambda z: attn(z, encoder_hidden_states=encoder_hidden_states, cross_attention_kwargs=cross_kw).sample,
                hidden_states)
        mm = self.motion_modules[i] if len(self.motion_modules) else None
        if mm is not None:
            hidden_states = mm(hidden_states, encoder_hidden_states=encoder_hidden_states,
                               cross_attention_kwargs=motion_kw)
        return hidden_states

    def _scales(self, cross_kw, motion_kw):
        """`lora_scale` / `motion_lora_scale` attributes (unet_blocks.py:367-375)."""
        ls = getattr(self, "lora_scale", None)
        if ls is not None and cross_kw is not None:
            cross_kw["scale"] = ls
        ms = getattr(self, "motion_lora_scale", None)
        if ms is not None:
            motion_kw = {"scale": ms} if motion_kw is None else {**motion_kw, "scale": ms}
        return cross_kw, motion_kw


class _DownBase(_Block3D):
    def __init__(self, in_channels, out_channels, temb_channels, num_layers, resnet_eps, resnet_act_fn,
                 resnet_groups, add_downsample, downsample_padding, cross, attn_num_head_channels,
                 cross_attention_dim, use_motion_module, motion_module_type, motion_module_kwargs):
        super().__init__()
        self.has_cross_attention = cross
        self.resnets = nn.ModuleList([
            _resnet(in_channels if i == 0 else out_channels, out_channels, temb_channels, resnet_eps,
                    resnet_groups, resnet_act_fn) for i in range(num_layers)])
        if cross:
            self.attentions = nn.ModuleList([
                _transformer(attn_num_head_channels, out_channels, cross_attention_dim, resnet_groups)
                for _ in range(num_layers)])
        mms = [get_motion_module(out_channels, motion_module_type, motion_module_kwargs) if use_motion_module else None
               for _ in range(num_layers)]
        self.motion_modules = nn.ModuleList(mms) if use_motion_module else mms
        self.downsamplers = (nn.ModuleList([D.Downsample2D(out_channels, use_conv=True, out_channels=out_channels,
                                                           padding=downsample_padding, name="op")])
                             if add_downsample else None)
        self.gradient_checkpointing = False

    def _run(self, hidden_states, temb, encoder_hidden_states, cross_kw, motion_kw, traj_features):
        if self.training and self.gradient_checkpointing:
            raise NotImplementedError          # unet_blocks.py:378-379
        f = hidden_states.shape[2]
        temb_rep = repeat(temb, "b c -> (b f) c", f=f)
        cross_kw, motion_kw = self._scales(cross_kw, motion_kw)
        outs = ()
        for i in range(len(self.resnets)):
            hidden_states = self._layer(i, hidden_states, temb_rep, encoder_hidden_states, cross_kw, motion_kw)
            outs += (hidden_states,)
        if traj_features is not None:          # modified_modules.py:115-117 / :172-174
            hidden_states = hidden_states + traj_features[self.traj_fea_idx]
            outs = outs[:-1] + (hidden_states,)
        if self.downsamplers is not None:
            for ds in self.downsamplers:
                hidden_states = _per_frame(ds, hidden_states)
            outs += (hidden_states,)
        return hidden_states, outs


class CrossAttnDownBlock3D(_DownBase):
    """unet_blocks.py:268-426 with the OMC patch of modified_modules.py:52-127 folded in:
    `traj_features` is popped from `cross_attention_kwargs` before it can reach diffusers."""

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6,
                 resnet_act_fn="swish", resnet_groups=32, attn_num_head_channels=1, cross_attention_dim=1280,
                 downsample_padding=1, add_downsample=True, use_motion_module=None, motion_module_type=None,
                 motion_module_kwargs=None, **_ignored):
        super().__init__(in_channels, out_channels, temb_channels, num_layers, resnet_eps, resnet_act_fn,
                         resnet_groups, add_downsample, downsample_padding, True, attn_num_head_channels,
                         cross_attention_dim, use_motion_module, motion_module_type, motion_module_kwargs)
        self.attn_num_head_channels = attn_num_head_channels

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None,
                motion_module_alpha=1.0, cross_attention_kwargs=None, motion_cross_attention_kwargs=None):
        cross_kw = dict(cross_attention_kwargs or {})
        traj = cross_kw.pop("traj_features", None)
        return self._run(hidden_states, temb, encoder_hidden_states, cross_kw,
                         dict(motion_cross_attention_kwargs or {}), traj)


class DownBlock3D(_DownBase):
    """unet_blocks.py:429-540 + modified_modules.py:129-185 (`traj_features` arrives via **kwargs,
    which the U-Net never fills for this block: unet_cam_obj.py:1227-1234)."""

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6,
                 resnet_act_fn="swish", resnet_groups=32, add_downsample=True, downsample_padding=1,
                 use_motion_module=None, motion_module_type=None, motion_module_kwargs=None, **_ignored):
        super().__init__(in_channels, out_channels, temb_channels, num_layers, resnet_eps, resnet_act_fn,
                         resnet_groups, add_downsample, downsample_padding, False, None, None,
                         use_motion_module, motion_module_type, motion_module_kwargs)

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, motion_module_alpha=1.0,
                motion_cross_attention_kwargs=None, **kwargs):
        traj = kwargs.pop("traj_features", None)
        return self._run(hidden_states, temb, encoder_hidden_states, None,
                         dict(motion_cross_attention_kwargs or {}), traj)


class UNetMidBlock3DCrossAttn(_Block3D):
    """unet_blocks.py:144-265: resnet0, then per layer (transformer, [motion], resnet)."""

    has_cross_attention = True

    def __init__(self, in_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_act_fn="swish",
                 resnet_groups=32, attn_num_head_channels=1, output_scale_factor=1.0, cross_attention_dim=1280,
                 use_motion_module=None, motion_module_type=None, motion_module_kwargs=None, **_ignored):
        super().__init__()
        self.attn_num_head_channels = attn_num_head_channels
        self.resnets = nn.ModuleList([
            _resnet(in_channels, in_channels, temb_channels, resnet_eps, resnet_groups, resnet_act_fn,
                    output_scale_factor) for _ in range(num_layers + 1)])
        self.attentions = nn.ModuleList([
            _transformer(attn_num_head_channels, in_channels, cross_attention_dim, resnet_groups)
            for _ in range(num_layers)])
        mms = [get_motion_module(in_channels, motion_module_type, motion_module_kwargs) if use_motion_module else None
               for _ in range(num_layers)]
        self.motion_modules = nn.ModuleList(mms) if use_motion_module else mms

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None,
                motion_module_alpha=1.0, cross_attention_kwargs=None, motion_cross_attention_kwargs=None):
        f = hidden_states.shape[2]
        temb_rep = repeat(temb, "b c -> (b f) c", f=f)
        cross_kw = cross_attention_kwargs
        ls = getattr(self, "lora_scale", None)
        if ls is not None:
            cross_kw = {"scale": ls}            # unet_blocks.py:240-242 replaces the dict
        motion_kw = motion_cross_attention_kwargs
        ms = getattr(self, "motion_lora_scale", None)
        if ms is not None:
            motion_kw = {"scale": ms} if motion_kw is None else {**motion_kw, "scale": ms}
        hidden_states = _per_frame(self.resnets[0], hidden_states, temb_rep)
        for i, attn in enumerate(self.attentions):
            hidden_states = _per_frame(
                lambda z: attn(z, encoder_hidden_states=encoder_hidden_states, cross_attention_kwargs=cross_kw).sample,
                hidden_states)
            mm = self.motion_modules[i] if len(self.motion_modules) else None
            if mm is not None:
                hidden_states = mm(hidden_states, encoder_hidden_states=encoder_hidden_states,
                                   cross_attention_kwargs=motion_kw)
            hidden_states = _per_frame(self.resnets[i + 1], hidden_states, temb_rep)
        return hidden_states


class _UpBase(_Block3D):
    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers, resnet_eps,
                 resnet_act_fn, resnet_groups, add_upsample, cross, attn_num_head_channels, cross_attention_dim,
                 use_motion_module, motion_module_type, motion_module_kwargs):
        super().__init__()
        self.has_cross_attention = cross
        resnets = []
        for i in range(num_layers):             # channel arithmetic: unet_blocks.py:579-580 / :735-736
            skip = in_channels if i == num_layers - 1 else out_channels
            cin = prev_output_channel if i == 0 else out_channels
            resnets.append(_resnet(cin + skip, out_channels, temb_channels, resnet_eps, resnet_groups, resnet_act_fn))
        self.resnets = nn.ModuleList(resnets)
        if cross:
            self.attentions = nn.ModuleList([
                _transformer(attn_num_head_channels, out_channels, cross_attention_dim, resnet_groups)
                for _ in range(num_layers)])
        mms = [get_motion_module(out_channels, motion_module_type, motion_module_kwargs) if use_motion_module else None
               for _ in range(num_layers)]
        self.motion_modules = nn.ModuleList(mms) if use_motion_module else mms
        self.upsamplers = (nn.ModuleList([D.Upsample2D(out_channels, use_conv=True, out_channels=out_channels)])
                           if add_upsample else None)
        self.gradient_checkpointing = False

    def _run(self, hidden_states, res_hidden_states_tuple, temb, encoder_hidden_states, upsample_size,
             cross_kw, motion_kw):
        if self.training and self.gradient_checkpointing:
            raise NotImplementedError
        f = hidden_states.shape[2]
        temb_rep = repeat(temb, "b c -> (b f) c", f=f)
        for i in range(len(self.resnets)):
            skip = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = torch.cat([hidden_states, skip], dim=1)
            hidden_states = self._layer(i, hidden_states, temb_rep, encoder_hidden_states, cross_kw, motion_kw)
        if self.upsamplers is not None:
            for up in self.upsamplers:
                hidden_states = _per_frame(up, hidden_states, upsample_size)
        return hidden_states


class CrossAttnUpBlock3D(_UpBase):
    """unet_blocks.py:543-706."""

    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers=1,
                 resnet_eps=1e-6, resnet_act_fn="swish", resnet_groups=32, attn_num_head_channels=1,
                 cross_attention_dim=1280, add_upsample=True, use_motion_module=None, motion_module_type=None,
                 motion_module_kwargs=None, **_ignored):
        super().__init__(in_channels, prev_output_channel, out_channels, temb_channels, num_layers, resnet_eps,
                         resnet_act_fn, resnet_groups, add_upsample, True, attn_num_head_channels,
                         cross_attention_dim, use_motion_module, motion_module_type, motion_module_kwargs)
        self.attn_num_head_channels = attn_num_head_channels

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None,
                upsample_size=None, attention_mask=None, motion_module_alpha=1.0, cross_attention_kwargs=None,
                motion_cross_attention_kwargs=None):
        cross_kw = cross_attention_kwargs
        ls = getattr(self, "lora_scale", None)
        if ls is not None:
            cross_kw = {"scale": ls}            # unet_blocks.py:645-647 replaces the dict
        motion_kw = dict(motion_cross_attention_kwargs or {})
        ms = getattr(self, "motion_lora_scale", None)
        if ms is not None:
            motion_kw["scale"] = ms
        return self._run(hidden_states, res_hidden_states_tuple, temb, encoder_hidden_states, upsample_size,
                         cross_kw, motion_kw)


class UpBlock3D(_UpBase):
    """unet_blocks.py:709-817."""

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers=1,
                 resnet_eps=1e-6, resnet_act_fn="swish", resnet_groups=32, add_upsample=True,
                 use_motion_module=None, motion_module_type=None, motion_module_kwargs=None, **_ignored):
        super().__init__(in_channels, prev_output_channel, out_channels, temb_channels, num_layers, resnet_eps,
                         resnet_act_fn, resnet_groups, add_upsample, False, None, None,
                         use_motion_module, motion_module_type, motion_module_kwargs)

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, upsample_size=None,
                encoder_hidden_states=None, motion_module_alpha=1.0, motion_cross_attention_kwargs=None, **kwargs):
        motion_kw = dict(motion_cross_attention_kwargs or {})
        ms = getattr(self, "motion_lora_scale", None)
        if ms is not None:
            motion_kw["scale"] = ms
        return self._run(hidden_states, res_hidden_states_tuple, temb, encoder_hidden_states, upsample_size,
                         None, motion_kw)


# ----------------------------------------------------------------------------
# the U-Net  (fmc/models/unet_cam_obj.py == fmc/models/unet.py + traj_features)
# ----------------------------------------------------------------------------
SD15_UNET_CONFIG = dict(
    sample_size=64, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
    mid_block_type="UNetMidBlock3DCrossAttn",
    up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
    block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1,
    mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=768,
    attention_head_dim=8,
)

_DOWN = {"CrossAttnDownBlock3D": CrossAttnDownBlock3D, "DownBlock3D": DownBlock3D}
_UP = {"CrossAttnUpBlock3D": CrossAttnUpBlock3D, "UpBlock3D": UpBlock3D}


class _Config(dict):
    __getattr__ = dict.__getitem__


class UNet3DConditionOutput:
    def __init__(self, sample):
        self.sample = sample


class UNet3DConditionModelCamObjCond(nn.Module):
    """`UNet3DConditionModel` (unet_cam_obj.py:49-826) + the pose/traj conditioned forward
    (:1107-1375).  With `pose_embedding_features=None` and plain `AttnProcessor`s it is the
    unconditioned base U-Net (unet.py:539-760) used by BASELINE config 1."""

    def __init__(self, sample_size=None, in_channels=4, out_channels=4, center_input_sample=False,
                 flip_sin_to_cos=True, freq_shift=0,
                 down_block_types=SD15_UNET_CONFIG["down_block_types"], mid_block_type="UNetMidBlock3DCrossAttn",
                 up_block_types=SD15_UNET_CONFIG["up_block_types"], only_cross_attention=False,
                 block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1,
                 mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5,
                 cross_attention_dim=1280, attention_head_dim=8, use_motion_module=False,
                 motion_module_resolutions=(1, 2, 4, 8), motion_module_mid_block=False, motion_module_type=None,
                 motion_module_kwargs=None, decoder_add_posecond=True, **_unused):
        super().__init__()
        motion_module_kwargs = dict(motion_module_kwargs or {})
        self.config = _Config(in_channels=in_channels, out_channels=out_channels, sample_size=sample_size,
                              center_input_sample=center_input_sample, block_out_channels=tuple(block_out_channels),
                              cross_attention_dim=cross_attention_dim, attention_head_dim=attention_head_dim,
                              layers_per_block=layers_per_block)
        self.decoder_add_posecond = decoder_add_posecond
        self.sample_size = sample_size
        boc = list(block_out_channels)
        n = len(boc)
        temb_dim = boc[0] * 4
        heads = (attention_head_dim,) * n if isinstance(attention_head_dim, int) else tuple(attention_head_dim)

        self.conv_in = InflatedConv3d(in_channels, boc[0], kernel_size=3, padding=(1, 1))
        self.time_proj = D.Timesteps(boc[0], flip_sin_to_cos, freq_shift)
        self.time_embedding = D.TimestepEmbedding(boc[0], temb_dim)

        common = dict(temb_channels=temb_dim, resnet_eps=norm_eps, resnet_act_fn=act_fn,
                      resnet_groups=norm_num_groups, cross_attention_dim=cross_attention_dim,
                      motion_module_type=motion_module_type, motion_module_kwargs=motion_module_kwargs)
        self.down_blocks = nn.ModuleList()
        cout = boc[0]
        for i, kind in enumerate(down_block_types):
            cin, cout = cout, boc[i]
            self.down_blocks.append(_DOWN[kind](
                in_channels=cin, out_channels=cout, num_layers=layers_per_block, add_downsample=i != n - 1,
                attn_num_head_channels=heads[i], downsample_padding=downsample_padding,
                use_motion_module=use_motion_module and (2 ** i in motion_module_resolutions), **common))
        assert mid_block_type == "UNetMidBlock3DCrossAttn"
        self.mid_block = UNetMidBlock3DCrossAttn(
            in_channels=boc[-1], output_scale_factor=mid_block_scale_factor, attn_num_head_channels=heads[-1],
            use_motion_module=use_motion_module and motion_module_mid_block, **common)
        self.up_blocks = nn.ModuleList()
        self.num_upsamplers = 0
        rev, rheads = boc[::-1], heads[::-1]
        cout = rev[0]
        for i, kind in enumerate(up_block_types):
            prev, cout = cout, rev[i]
            cin = rev[min(i + 1, n - 1)]
            last = i == n - 1
            self.num_upsamplers += 0 if last else 1
            self.up_blocks.append(_UP[kind](
                in_channels=cin, out_channels=cout, prev_output_channel=prev, num_layers=layers_per_block + 1,
                add_upsample=not last, attn_num_head_channels=rheads[i],
                use_motion_module=use_motion_module and (2 ** (3 - i) in motion_module_resolutions), **common))
        self.conv_norm_out = nn.GroupNorm(num_channels=boc[0], num_groups=norm_num_groups, eps=norm_eps)
        self.conv_act = nn.SiLU()
        self.conv_out = InflatedConv3d(boc[0], out_channels, kernel_size=3, padding=1)

    # -- dtype / device helpers of ModelMixin ---------------------------------
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def in_channels(self):
        return self.config.in_channels

    # -- processor registries (unet_cam_obj.py:322-468) ----------------------
    def _processors(self, temporal: bool):
        out = {}
        for name, mod in self.named_modules():
            if hasattr(mod, "set_processor") and (("motion_modules." in name) == temporal):
                out[f"{name}.processor"] = mod
        return out

    @property
    def attn_processors(self):
        return {k: m.processor for k, m in self._processors(False).items()}

    @property
    def mm_attn_processors(self):
        return {k: m.processor for k, m in self._processors(True).items()}

    def _set(self, temporal, processor):
        mods = self._processors(temporal)
        if isinstance(processor, dict) and len(processor) != len(mods):
            raise ValueError(f"A dict of processors was passed, but the number of processors {len(processor)} does "
                             f"not match the number of attention layers: {len(mods)}.")
        for k, m in mods.items():
            m.set_processor(processor.pop(k) if isinstance(processor, dict) else processor)

    def set_attn_processor(self, processor):
        self._set(False, processor)

    def set_mm_attn_processor(self, processor):
        self._set(True, processor)

    def _hidden_size(self, name):
        boc = self.config.block_out_channels
        if name.startswith("mid_block"):
            return boc[-1], -1, False
        idx = int(name.split(".")[1])
        if name.startswith("up_blocks"):
            return list(reversed(boc))[idx], idx, True
        return boc[idx], idx, False

    def set_all_attn_processor(self, add_spatial=False, spatial_attn_names="attn1", add_temporal=False,
                               add_spatial_lora=True, add_motion_lora=False, temporal_attn_names="0",
                               pose_feature_dimensions=(320, 640, 1280, 1280), lora_kwargs=None,
                               motion_lora_kwargs=None, **attention_processor_kwargs):
        """unet_cam_obj.py:983-1105: which processor class goes where."""
        lora_kwargs, motion_lora_kwargs = dict(lora_kwargs or {}), dict(motion_lora_kwargs or {})
        lora_rank = lora_kwargs.pop("lora_rank")
        motion_lora_rank = motion_lora_kwargs.pop("lora_rank")
        pfd = list(pose_feature_dimensions)

        def build(names, temporal, add_pose, add_lora, rank_cfg, sel_names, extra_lora_kw):
            procs = {}
            chosen = sel_names.split(",")
            for name in names:
                attn_name = name.split(".")[-2]
                hidden, idx, is_up = self._hidden_size(name)
                cross = None if (temporal or attn_name == "attn1") else self.config.cross_attention_dim
                rank = rank_cfg if rank_cfg > 16 else hidden // rank_cfg
                pose = add_pose and attn_name in chosen
                if pose and temporal and is_up:
                    pose = pose and self.decoder_add_posecond
                pdim = (list(reversed(pfd))[idx] if is_up else pfd[idx]) if pose else None
                if pose and add_lora:
                    procs[name] = LORAPoseAdaptorAttnProcessor(hidden_size=hidden, pose_feature_dim=pdim,
                                                               cross_attention_dim=cross, rank=rank,
                                                               **attention_processor_kwargs, **extra_lora_kw)
                elif pose:
                    procs[name] = PoseAdaptorAttnProcessor(hidden_size=hidden, pose_feature_dim=pdim,
                                                           cross_attention_dim=cross, **attention_processor_kwargs)
                elif add_lora:
                    procs[name] = LoRAAttnProcessor(hidden_size=hidden, cross_attention_dim=cross, rank=rank)
                else:
                    procs[name] = AttnProcessor()
            return procs

        self.set_attn_processor(build(list(self.attn_processors), False, add_spatial, add_spatial_lora,
                                      lora_rank, spatial_attn_names, lora_kwargs))
        self.set_mm_attn_processor(build(list(self.mm_attn_processors), True, add_temporal, add_motion_lora,
                                         motion_lora_rank, temporal_attn_names, motion_lora_kwargs))

    # -- forward (unet_cam_obj.py:1107-1375) ----------------------------------
    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, attention_mask=None,
                cross_attention_kwargs=None, pose_embedding_features: Optional[List[torch.Tensor]] = None,
                traj_features: Optional[List[torch.Tensor]] = None, return_dict: bool = True, **_unused):
        assert cross_attention_kwargs is None, \
            "the reference's `cross_attention_kwargs.update(...)` returns None (unet_cam_obj.py:1222)"
        up_factor = 2 ** self.num_upsamplers
        forward_upsample_size = any(s % up_factor != 0 for s in sample.shape[-2:])
        upsample_size = None
        if self.config.center_input_sample:
            sample = 2 * sample - 1.0
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.float64 if isinstance(t, float) else torch.int64, device=sample.device)
        elif t.ndim == 0:
            t = t[None].to(sample.device)
        t = t.expand(sample.shape[0])
        emb = self.time_embedding(self.time_proj(t).to(self.dtype))
        f = sample.shape[2]
        encoder_hidden_states = repeat(encoder_hidden_states, "b n c -> (b f) n c", f=f)
        sample = self.conv_in(sample)

        has_pose = pose_embedding_features is not None
        skips = (sample,)
        for i, blk in enumerate(self.down_blocks):
            pf = pose_embedding_features[i] if has_pose else None
            mkw = {"pose_feature": pf} if has_pose else {}
            if blk.has_cross_attention:
                ckw = {"pose_feature": pf, "traj_features": traj_features} if has_pose else {}
                sample, res = blk(hidden_states=sample, temb=emb, encoder_hidden_states=encoder_hidden_states,
                                  attention_mask=attention_mask, cross_attention_kwargs=ckw,
                                  motion_cross_attention_kwargs=mkw)
            else:
                sample, res = blk(hidden_states=sample, temb=emb, motion_cross_attention_kwargs=mkw)
            skips += res
        ckw = {"pose_feature": pose_embedding_features[-1]} if has_pose else None
        sample = self.mid_block(sample, emb, encoder_hidden_states=encoder_hidden_states,
                                attention_mask=attention_mask, cross_attention_kwargs=ckw,
                                motion_cross_attention_kwargs=ckw)
        for i, blk in enumerate(self.up_blocks):
            last = i == len(self.up_blocks) - 1
            n_res = len(blk.resnets)
            res, skips = skips[-n_res:], skips[:-n_res]
            if not last and forward_upsample_size:
                upsample_size = skips[-1].shape[-2:]      # spatial dims of the 5-D skip
            use_pose = has_pose and self.decoder_add_posecond
            pf = pose_embedding_features[-(i + 1)] if use_pose else None
            mkw = {"pose_feature": pf} if use_pose else {}
            if blk.has_cross_attention:
                sample = blk(hidden_states=sample, temb=emb, res_hidden_states_tuple=res,
                             encoder_hidden_states=encoder_hidden_states, upsample_size=upsample_size,
                             attention_mask=attention_mask,
                             cross_attention_kwargs={"pose_feature": pf} if use_pose else None,
                             motion_cross_attention_kwargs=mkw)
            else:
                sample = blk(hidden_states=sample, temb=emb, res_hidden_states_tuple=res,
                             upsample_size=upsample_size, motion_cross_attention_kwargs=mkw)
        sample = _per_frame(self.conv_norm_out, sample)
        sample = self.conv_out(self.conv_act(sample))
        return UNet3DConditionOutput(sample) if return_dict else (sample,)


# ----------------------------------------------------------------------------
# Camera Encoder  (fmc/models/pose_adaptor.py)
# ----------------------------------------------------------------------------
class _EncDownsample(nn.Module):
    """pose_adaptor.py:75-99 / adapter.py:35-61: stride-2 conv or 2x2 average pool."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels, self.out_channels, self.use_conv = channels, out_channels or channels, use_conv
        if use_conv:
            self.op = nn.Conv2d(channels, self.out_channels, 3, stride=2, padding=padding)
        else:
            assert self.channels == self.out_channels
            self.op = nn.AvgPool2d(kernel_size=2, stride=2)

    def forward(self, x):
        assert x.shape[1] == self.channels
        return self.op(x)


class _EncResnetBlock(nn.Module):
    """conv(in) -> 3x3 -> ReLU -> ksize conv, + skip.  `skep_on_input=True` is the camera
    encoder's variant (pose_adaptor.py:102-135, skip conv maps in_c), False is the OMC
    adapter's (adapter.py:64-98, skip conv maps out_c)."""

    def __init__(self, in_c, out_c, down, ksize=3, sk=False, use_conv=True, skep_on_input=True):
        super().__init__()
        in_c, out_c = int(in_c), int(out_c)
        ps = ksize // 2
        self.in_conv = nn.Conv2d(in_c, out_c, ksize, 1, ps) if (in_c != out_c or not sk) else None
        self.block1 = nn.Conv2d(out_c, out_c, 3, 1, 1)
        self.act = nn.ReLU()
        self.block2 = nn.Conv2d(out_c, out_c, ksize, 1, ps)
        self.skep = None if sk else nn.Conv2d(in_c if skep_on_input else out_c, out_c, ksize, 1, ps)
        self.down = down
        if down:
            self.down_opt = _EncDownsample(in_c, use_conv=use_conv)

    def forward(self, x):
        if self.down:
            x = self.down_opt(x)
        if self.in_conv is not None:
            x = self.in_conv(x)
        h = self.block2(self.act(self.block1(x)))
        return h + (self.skep(x) if self.skep is not None else x)


class CameraPoseEncoder(nn.Module):
    """pose_adaptor.py:159-240.  Per level `nums_rb` x [ResnetBlock -> `(b h w) f c` ->
    TemporalTransformerBlock (1 self-attention with PE, default processor) -> back];
    the tensor after each level is a feature `(b f) c h w`."""

    def __init__(self, downscale_factor, channels=(320, 640, 1280, 1280), nums_rb=3, cin=64, ksize=3, sk=False,
                 use_conv=True, compression_factor=1, temporal_attention_nhead=8,
                 attention_block_types=("Temporal_Self",), temporal_position_encoding=False,
                 temporal_position_encoding_max_len=16, rescale_output_factor=1.0):
        super().__init__()
        self.unshuffle = nn.PixelUnshuffle(downscale_factor)
        self.channels, self.nums_rb = list(channels), nums_rb
        self.encoder_down_conv_blocks = nn.ModuleList()
        self.encoder_down_attention_blocks = nn.ModuleList()
        for i, ch in enumerate(self.channels):
            convs, attns = nn.ModuleList(), nn.ModuleList()
            mid = int(ch / compression_factor)
            for j in range(nums_rb):
                if j == 0:
                    cin_j, cout_j, down = (self.channels[i - 1], mid, True) if i != 0 else (self.channels[0], mid, False)
                elif j == nums_rb - 1:
                    cin_j, cout_j, down = mid, ch, False
                else:
                    cin_j, cout_j, down = mid, mid, False
                convs.append(_EncResnetBlock(cin_j, cout_j, down=down, ksize=ksize, sk=sk, use_conv=use_conv))
                attns.append(TemporalTransformerBlock(
                    dim=cout_j, num_attention_heads=temporal_attention_nhead,
                    attention_head_dim=int(cout_j / temporal_attention_nhead),
                    attention_block_types=tuple(attention_block_types), dropout=0.0, cross_attention_dim=None,
                    temporal_position_encoding=temporal_position_encoding,
                    temporal_position_encoding_max_len=temporal_position_encoding_max_len,
                    rescale_output_factor=rescale_output_factor))
            self.encoder_down_conv_blocks.append(convs)
            self.encoder_down_attention_blocks.append(attns)
        self.encoder_conv_in = nn.Conv2d(cin, self.channels[0], 3, 1, 1)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def forward(self, x):
        bs = x.shape[0]
        x = self.unshuffle(rearrange(x, "b c f h w -> (b f) c h w"))
        x = self.encoder_conv_in(x)
        feats = []
        for convs, attns in zip(self.encoder_down_conv_blocks, self.encoder_down_attention_blocks):
            for conv, attn in zip(convs, attns):
                x = conv(x)
                h, w = x.shape[-2:]
                x = attn(rearrange(x, "(b f) c h w -> (b h w) f c", b=bs))
                x = rearrange(x, "(b h w) f c -> (b f) c h w", h=h, w=w)
            feats.append(x)
        return feats


class PoseAdaptor(nn.Module):
    """pose_adaptor.py:56-72."""

    def __init__(self, unet, pose_encoder):
        super().__init__()
        self.unet, self.pose_encoder = unet, pose_encoder

    def forward(self, noisy_latents, timesteps, encoder_hidden_states, pose_embedding):
        assert pose_embedding.ndim == 5
        bs = pose_embedding.shape[0]
        feats = [rearrange(x, "(b f) c h w -> b c f h w", b=bs) for x in self.pose_encoder(pose_embedding)]
        return self.unet(noisy_latents, timesteps, encoder_hidden_states, pose_embedding_features=feats).sample


class CamObjPoseAdaptor(nn.Module):
    """pose_obj_adaptor.py:7-23."""

    def __init__(self, unet, pose_encoder):
        super().__init__()
        self.unet, self.pose_encoder = unet, pose_encoder

    def forward(self, noisy_latents, timesteps, encoder_hidden_states, pose_embedding, traj_features):
        assert pose_embedding.ndim == 5
        bs = pose_embedding.shape[0]
        feats = [rearrange(x, "(b f) c h w -> b c f h w", b=bs) for x in self.pose_encoder(pose_embedding)]
        return self.unet(noisy_latents, timesteps, encoder_hidden_states, pose_embedding_features=feats,
                         traj_features=traj_features).sample


# ----------------------------------------------------------------------------
# Object Encoder  (fmc/adapter.py:109-192)
# ----------------------------------------------------------------------------
class Adapter(nn.Module):
    """T2I-Adapter style encoder.  After every level: zero 1x1 conv, then `x = nearest(mask) * x`
    with a *cascaded* mask pyramid (the mask is re-interpolated from the previous level's mask,
    adapter.py:175-177) and the masked tensor is both the emitted feature and the next level's input."""

    def __init__(self, channels=(320, 640, 1280, 1280), nums_rb=3, cin=64, ksize=3, sk=False, use_conv=True,
                 align_training_size=0, use_pre_zero_conv=False, use_post_zero_conv=False):
        super().__init__()
        assert align_training_size == 0
        self.align_training_size = align_training_size
        self.unshuffle = nn.PixelUnshuffle(8)
        self.channels, self.nums_rb = list(channels), nums_rb
        body = []
        for i, ch in enumerate(self.channels):
            for j in range(nums_rb):
                first_of_level = i != 0 and j == 0
                body.append(_EncResnetBlock(self.channels[i - 1] if first_of_level else ch, ch, down=first_of_level,
                                            ksize=ksize, sk=sk, use_conv=use_conv, skep_on_input=False))
        self.body = nn.ModuleList(body)
        self.conv_in = nn.Conv2d(cin, self.channels[0], 3, 1, 1)

        def zero_conv(c):
            m = nn.Conv2d(c, c, kernel_size=1, stride=1, padding=0)
            nn.init.zeros_(m.weight)
            nn.init.zeros_(m.bias)
            return m

        self.zero_conv_in = zero_conv(cin) if use_pre_zero_conv else nn.Identity()
        self.zero_conv_out_list = nn.ModuleList(
            [zero_conv(c) if use_post_zero_conv else nn.Identity() for c in self.channels])

    def forward(self, x, mask_feat):
        x = self.conv_in(self.zero_conv_in(self.unshuffle(x)))
        feats = []
        for i in range(len(self.channels)):
            for j in range(self.nums_rb):
                x = self.body[i * self.nums_rb + j](x)
            x = self.zero_conv_out_list[i](x)
            if mask_feat is not None:
                mask_feat = F.interpolate(mask_feat, size=x.shape[-2:], mode="nearest")
                x = mask_feat * x
            feats.append(x)
        return feats


def patch_down_blocks_for_omc(unet: nn.Module) -> None:
    """What train_cam_obj_ctrl.py:317-329 does: tag every down block with `traj_fea_idx`
    (the oracle blocks already contain the patched forward)."""
    idx = 0
    for name, m in unet.down_blocks.named_modules():
        if m.__class__.__name__ in ("CrossAttnDownBlock3D", "DownBlock3D"):
            m.traj_fea_idx = idx
            idx += 1
