"""Attention processors of FMC on the gfx950 kernels.

Mirrors `fmc/models/attention_processor.py` (same four classes, constructor arguments, parameter names
`qkv_merge` / `q_merge` / `kv_merge` / `to_{q,k,v,out}_lora.{down,up}` and call signatures, so they install
through `set_attn_processor` / `set_mm_attn_processor` and load the reference's checkpoints), but the chain
`head_to_batch_dim -> baddbmm -> softmax -> bmm -> batch_to_head_dim` (reference :61-67, :148-154, :271-281,
:402-408) collapses into one `fmc_spatial_attn_fwd` / `fmc_temporal_attn_fwd` launch, Q/K/V come from ONE fused
projection GEMM, and a frozen LoRA is merged into the projection weights (`W + s * up @ down`).

Spatial vs temporal is decided by the shape of `hidden_states`:
  * `[B, S, C]` tokens: attention over S (spatial self / text cross attention);
  * with `temporal=True` (set by `TemporalSelfAttention`): `[B, F, P, C]` native channels-last video
    tokens or `[N, F, C]` reference-layout tokens: attention over F.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from .. import hip_ops as K
from .layers import LoRALinearLayer, linear_op


_POSE_TERM = os.environ.get("FMC_NO_POSE_TERM", "0") != "1"      # A/B switch for the pre-computed Camera-Adapter term
_MERGE_FOLD = os.environ.get("FMC_MERGE_FOLD", "1") != "0"       # A/B switch: the Camera-Adapter merge folded into the q | k | v projection (un-fused chain)
MERGE_FOLD_MIN_DIM = 1280                                        # ... at the levels where it was measured to pay (C = 1280; tests lower it to cover the path at small widths)


def _tok(x: torch.Tensor) -> torch.Tensor:
    """`b c h w -> b (h w) c` (free on channels-last storage); token tensors pass through."""
    if x.ndim == 4:
        n, c, h, w = x.shape
        t = x.permute(0, 2, 3, 1)
        return (t if t.is_contiguous() else t.contiguous()).view(n, h * w, c)
    return x


def _attention_core(attn, q_in: torch.Tensor, kv_in: Optional[torch.Tensor], temporal: bool, lora=None,
                    lora_scale: float = 1.0, residual: Optional[torch.Tensor] = None, text_context: bool = False, qkv_fold=None) -> torch.Tensor:
    """Fused projection -> attention kernel -> output projection (bias, dropout p=0) [+ residual].  `qkv_fold` = (W', term) of
    `_PoseMerge._qkv_fold`: q | k | v = `q_in W'^T + term` (q_in = the UN-merged tokens)."""
    heads = attn.heads
    w_a, w_b, w_o = attn.fused_weights(lora, lora_scale)
    if qkv_fold is not None:
        assert kv_in is None
        o = K.self_attention_qkv(linear_op(q_in, qkv_fold[0], None, qkv_fold[1]), heads, attn.scale, temporal)
        return linear_op(o, w_o, attn.to_out[0].bias, residual, ln=attn.__dict__.get("_next_ln") if residual is not None else None,
                         lazy_residual=residual is not None and bool(attn.__dict__.get("_lazy_res")))
    c = attn.inner_dim
    fp8 = attn.__dict__.get("_fp8_scales") if (temporal and kv_in is None) else None
    if (fp8 is not None and q_in.is_cuda and q_in.dtype == torch.bfloat16 and w_a.dtype == torch.bfloat16
            and q_in.is_contiguous() and c % 64 == 0 and q_in.shape[-3 if q_in.ndim == 4 else -2] in (16, 32)):
        # fp8 temporal attention (BASELINE configs[4]): the projection's epilogue emits e4m3 q | k | v with per-tensor
        # scales, QK^T runs on the fp8 MFMA -- one autograd node from the normed tokens to the attention output
        o = K.temporal_attention_fp8(q_in, w_a if w_a.is_contiguous() else w_a.contiguous(), fp8, heads, attn.scale)
    elif kv_in is None:                                 # self attention: one [.., 3C] GEMM
        qkv = linear_op(q_in, w_a)
        o = K.self_attention_qkv(qkv, heads, attn.scale, temporal)      # q | k | v stay slices of the fused output
    elif w_b is None:
        # a SELF-attention module whose query and key / value inputs differ: the q-only / kv-only Camera-Adapter merge (reference
        # attention_processor.py:193-200, :259-265 -- `q_merge` / `kv_merge`; un-shipped configurations).  The fused [3C, C] weight is used as its
        # q rows and its k | v rows on the two inputs; the results are laid side by side so that the fused-QKV attention front-end (and its
        # backward) serves spatial and temporal tokens alike
        C = attn.inner_dim
        assert q_in.shape == kv_in.shape, "q-only / kv-only pose merge: query and key / value tokens must have the same shape"
        qkv = torch.cat([linear_op(q_in, w_a[:C]), linear_op(kv_in, w_a[C:])], dim=-1)
        o = K.self_attention_qkv(qkv, heads, attn.scale, temporal)
    else:                                               # cross attention: q GEMM + one [.., 2C] kv GEMM
        assert not temporal
        q = linear_op(q_in, w_a)
        # (text cross attention: k | v of the constant text once per clip -- `Attention.text_kv`; a pose-merged context changes every call)
        kv = attn.text_kv(kv_in, w_b)[0] if text_context else linear_op(kv_in, w_b)
        o = K.cross_attention_q_kv(q, kv, heads, attn.scale)
    # (`_next_ln`: the LayerNorm the block applies to `attn(x) + x` -- set by the block, consumed by hip_ops.linear when the tile holds whole rows)
    return linear_op(o, w_o, attn.to_out[0].bias, residual, ln=attn.__dict__.get("_next_ln") if residual is not None else None,
                     lazy_residual=residual is not None and bool(attn.__dict__.get("_lazy_res")))


def _fusable(attn, residual, shape4):
    """The block-level `attn(x) + hidden_states` may ride in the output projection only when the processor applies
    nothing after it (`residual_connection` off, `rescale_output_factor` 1) and the tokens are not re-shaped."""
    if residual is None:
        return None
    assert shape4 is None and not attn.residual_connection and attn.rescale_output_factor == 1.0, \
        "internal: fused block residual needs a plain attention module"
    return residual


def _finish(attn, out, residual, shape4):
    if shape4 is not None:
        n, c, h, w = shape4
        out = out.view(n, h, w, c).permute(0, 3, 1, 2)
    if attn.residual_connection:
        out = out + residual
    if attn.rescale_output_factor != 1.0:
        out = out / attn.rescale_output_factor
    return out


class AttnProcessor:
    """Reference: attention_processor.py:15-82 (`pose_feature` accepted and ignored, :28)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 scale: float = 1.0, pose_feature=None, temporal: bool = False, _residual=None):
        attn.prepare_attention_mask(attention_mask, 0, 0)
        shape4 = hidden_states.shape if (hidden_states.ndim == 4 and not temporal) else None
        x = _tok(hidden_states) if not temporal else hidden_states
        out = _attention_core(attn, x, encoder_hidden_states, temporal, residual=_fusable(attn, _residual, shape4),
                              text_context=encoder_hidden_states is not None)
        return _finish(attn, out, hidden_states, shape4)


class LoRAAttnProcessor(nn.Module):
    """Reference: attention_processor.py:85-169 (`W x + s * up(down(x))` on all four projections)."""

    def __init__(self, hidden_size=None, cross_attention_dim=None, rank=4, network_alpha=None, lora_scale=1.0):
        super().__init__()
        self.rank, self.lora_scale = rank, lora_scale
        kv_in = cross_attention_dim or hidden_size
        self.to_q_lora = LoRALinearLayer(hidden_size, hidden_size, rank, network_alpha)
        self.to_k_lora = LoRALinearLayer(kv_in, hidden_size, rank, network_alpha)
        self.to_v_lora = LoRALinearLayer(kv_in, hidden_size, rank, network_alpha)
        self.to_out_lora = LoRALinearLayer(hidden_size, hidden_size, rank, network_alpha)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 pose_feature=None, scale=None, temporal: bool = False, _residual=None):
        _require_frozen(self)
        attn.prepare_attention_mask(attention_mask, 0, 0)
        s = resolve_lora_scale(self, scale)
        shape4 = hidden_states.shape if (hidden_states.ndim == 4 and not temporal) else None
        x = _tok(hidden_states) if not temporal else hidden_states
        out = _attention_core(attn, x, encoder_hidden_states, temporal, lora=self, lora_scale=s,
                              residual=_fusable(attn, _residual, shape4), text_context=encoder_hidden_states is not None)
        return _finish(attn, out, hidden_states, shape4)


_NO_SCALE = object()


def resolve_lora_scale(proc, scale=_NO_SCALE) -> float:
    """The LoRA scale a processor call resolves to -- ONE rule for the un-fused chain and the fused temporal block.
    `LoRAAttnProcessor.__call__(..., scale=None)` (reference :108-116): `lora_scale` unless a scale is passed;
    `LORAPoseAdaptorAttnProcessor.__call__(..., scale=1.0)` (reference :337-347): 1.0 unless a scale is passed, `lora_scale` only for an
    explicit `scale=None`."""
    if scale is _NO_SCALE:
        return 1.0 if isinstance(proc, LORAPoseAdaptorAttnProcessor) else proc.lora_scale
    return proc.lora_scale if scale is None else scale


def _require_frozen(proc: nn.Module) -> None:
    if torch.is_grad_enabled() and any(p.requires_grad for n, p in proc.named_parameters() if "_lora" in n):
        raise NotImplementedError(
            "the gfx950 path merges LoRA into the projection weights; training the LoRA itself (FMC stage 1, a 2-D "
            "U-Net, out of scope) is not supported -- call `.requires_grad_(False)` on the LoRA layers")


class _PoseMerge:
    def _build_merge(self, hidden_size, pose_feature_dim, query_condition, key_value_condition):
        assert hidden_size == pose_feature_dim
        self.query_condition, self.key_value_condition = query_condition, key_value_condition
        name = "qkv_merge" if (query_condition and key_value_condition) else ("q_merge" if query_condition else "kv_merge")
        layer = nn.Linear(hidden_size, hidden_size)
        nn.init.zeros_(layer.weight)
        nn.init.zeros_(layer.bias)
        setattr(self, name, layer)

    def _merge(self, hidden_states, encoder_hidden_states, pose_feature, s):
        """`merge(h + pose) * s + h` (attention_processor.py:256-265)."""
        if self.query_condition and self.key_value_condition:
            w, b = self.qkv_merge.weight, self.qkv_merge.bias
            if (_POSE_TERM and not torch.is_grad_enabled() and hidden_states.is_cuda and hidden_states.dtype == torch.bfloat16
                    and w.dtype == torch.bfloat16 and pose_feature.shape == hidden_states.shape
                    and pose_feature.dtype == torch.bfloat16 and hidden_states.is_contiguous()):
                # merge is linear in pose and pose is constant over the denoising steps of a clip: its term
                # s * (W pose + b) is computed once and rides in the GEMM epilogue as a second residual, so the per-step
                # `h + pose` pass (one read of h and pose, one write, per motion module) is gone
                return K.linear(hidden_states, w, None, hidden_states, s, residual2=self._pose_term(pose_feature, w, b, s)), None
            m = linear_op(hidden_states + pose_feature, w, b, hidden_states, s)
            return m, None
        if self.query_condition:
            return torch.add(hidden_states, self.q_merge(hidden_states + pose_feature), alpha=s), encoder_hidden_states
        return hidden_states, torch.add(encoder_hidden_states, self.kv_merge(encoder_hidden_states + pose_feature), alpha=s)


def _pose_term_impl(self, pose_feature, w, b, s):
    """`s * (pose @ W^T + b)`, cached per processor.  The entry keeps `pose_feature` alive, so an equal data pointer
    means the same storage, and in-place writes to it bump the version counter it is keyed on."""
    key = (pose_feature.data_ptr(), pose_feature._version, tuple(pose_feature.shape), w.data_ptr(), w._version,
           None if b is None else b._version, float(s))
    hit = self.__dict__.get("_pose_term_cache")
    if hit is None or hit[0] != key:
        pf = pose_feature if pose_feature.is_contiguous() else pose_feature.contiguous()
        hit = (key, K.linear(pf, w, b, None, s), pose_feature)
        self.__dict__["_pose_term_cache"] = hit
    return hit[1]


_PoseMerge._pose_term = _pose_term_impl


def _qkv_fold_impl(self, attn, x, pose_feature, s, lora=None, lora_scale: float = 1.0):
    """The `qkv_merge` Camera-Adapter merge folded into the fused q | k | v projection of the un-fused chain (the 10x16 / 5x8 levels, C = 1280;
    reference attention_processor.py:255-283: `m = s (W_m (h + pose) + b_m) + h`, then `to_q / to_k / to_v (m)`).  Both maps are linear and nothing
    reads m but the three projections, so
        q | k | v = W_qkv (s W_m + I) h + W_qkv (s (W_m pose + b_m)) = W' h + term:
    W' [3C, C] once per set of weights (fp32 product, rounded once), `term` [.., 3C] once per clip from the cached pose term -- the merge GEMM (M x C x C
    per block and step) and the round trip of m are gone: 96.9 -> 62.1 us at M = 5120, 47.5 -> 22.0 us at M = 1280 (tools/scratch/r05/bench_fold.py).
    Returns (W', term) or None where the plain chain must run (training, fp32, fp8 projections, a layout mismatch, FMC_MERGE_FOLD=0).  The entry keeps
    the tensors the graph runner refreshes in place for a new clip (`_GraphedUNet.set_conditioning`)."""
    w, b = self.qkv_merge.weight, self.qkv_merge.bias
    if not (_MERGE_FOLD and _POSE_TERM and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
            and pose_feature.shape == x.shape and pose_feature.dtype == torch.bfloat16 and x.is_contiguous()
            and attn.inner_dim >= MERGE_FOLD_MIN_DIM and attn.__dict__.get("_fp8_scales") is None):
        return None
    w_a = attn.fused_weights(lora, lora_scale)[0]
    if w_a.dtype != torch.bfloat16 or w_a.shape[1] != w.shape[0]:
        return None
    term_src = self._pose_term(pose_feature, w, b, s)
    key = (w_a.data_ptr(), w_a._version, w.data_ptr(), w._version, float(s), term_src.data_ptr(), term_src._version, tuple(term_src.shape))
    hit = self.__dict__.get("_qkv_fold_cache")
    if hit is None or hit[0][:5] != key[:5]:
        eye = torch.eye(w.shape[0], device=w.device, dtype=torch.float32)
        w_f = (w_a.float() @ (float(s) * w.float() + eye)).to(torch.bfloat16).contiguous()
        hit = None
    else:
        w_f = hit[1]
    if hit is None or hit[0] != key:
        wq = w_a if w_a.is_contiguous() else w_a.contiguous()
        hit = (key, w_f, K.linear(term_src, wq), wq, term_src)
        self.__dict__["_qkv_fold_cache"] = hit
    return hit[1], hit[2]


_PoseMerge._qkv_fold = _qkv_fold_impl


def _pose_tokens(pose_feature, like):
    """Bring the pose feature to the token layout of `like` (`[B,F,P,C]`, `[N,F,C]` or `[B,S,C]`)."""
    if pose_feature.ndim == like.ndim and pose_feature.shape == like.shape:
        return pose_feature
    if pose_feature.ndim == 5:                     # b c f h w -> [B, F, (h w), C]; free on channels_last_3d storage
        b, c, f, h, w = pose_feature.shape
        t = pose_feature.permute(0, 2, 3, 4, 1)
        t = (t if t.is_contiguous() else t.contiguous()).view(b, f, h * w, c)
        if like.ndim == 3:                         # reference temporal layout (b h w) f c
            t = t.permute(0, 2, 1, 3).reshape(b * h * w, f, c)
        return t
    if pose_feature.ndim == 4:
        return _tok(pose_feature)
    return pose_feature


class PoseAdaptorAttnProcessor(nn.Module, _PoseMerge):
    """Camera Adapter.  Reference: attention_processor.py:172-293 (zero-initialised merge layer, `pose_feature`
    is the third positional argument of `forward`)."""

    def __init__(self, hidden_size, pose_feature_dim=None, cross_attention_dim=None, query_condition=False,
                 key_value_condition=False, scale=1.0):
        super().__init__()
        self.hidden_size, self.pose_feature_dim = hidden_size, pose_feature_dim
        self.cross_attention_dim, self.scale = cross_attention_dim, scale
        self._build_merge(hidden_size, pose_feature_dim, query_condition, key_value_condition)

    def forward(self, attn, hidden_states, pose_feature, encoder_hidden_states=None, attention_mask=None, temb=None,
                scale=None, temporal: bool = False, _residual=None):
        assert pose_feature is not None
        s = scale or self.scale
        attn.prepare_attention_mask(attention_mask, 0, 0)
        shape4 = hidden_states.shape if (hidden_states.ndim == 4 and not temporal) else None
        x = hidden_states if temporal else _tok(hidden_states)
        if self.query_condition and self.key_value_condition:
            assert encoder_hidden_states is None
        ctx = x if encoder_hidden_states is None else _tok(encoder_hidden_states)
        pose = _pose_tokens(pose_feature, x)
        fold = self._qkv_fold(attn, x, pose, s) if (self.query_condition and self.key_value_condition) else None
        if fold is not None:
            out = _attention_core(attn, x, None, temporal, residual=_fusable(attn, _residual, shape4), qkv_fold=fold)
            return _finish(attn, out, hidden_states, shape4)
        q_in, kv_in = self._merge(x, ctx, pose, s)
        out = _attention_core(attn, q_in, kv_in, temporal, residual=_fusable(attn, _residual, shape4))
        return _finish(attn, out, hidden_states, shape4)


class LORAPoseAdaptorAttnProcessor(nn.Module, _PoseMerge):
    """Reference: attention_processor.py:296-420 (pose merge with `self.scale` + LoRA projections)."""

    def __init__(self, hidden_size, pose_feature_dim=None, cross_attention_dim=None, query_condition=False,
                 key_value_condition=False, scale=1.0, rank=4, network_alpha=None, lora_scale=1.0):
        super().__init__()
        self.hidden_size, self.pose_feature_dim = hidden_size, pose_feature_dim
        self.cross_attention_dim, self.scale = cross_attention_dim, scale
        self._build_merge(hidden_size, pose_feature_dim, query_condition, key_value_condition)
        self.rank, self.lora_scale = rank, lora_scale
        kv_in = cross_attention_dim or hidden_size
        self.to_q_lora = LoRALinearLayer(hidden_size, hidden_size, rank, network_alpha)
        self.to_k_lora = LoRALinearLayer(kv_in, hidden_size, rank, network_alpha)
        self.to_v_lora = LoRALinearLayer(kv_in, hidden_size, rank, network_alpha)
        self.to_out_lora = LoRALinearLayer(hidden_size, hidden_size, rank, network_alpha)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0,
                 pose_feature=None, temporal: bool = False, _residual=None):
        assert pose_feature is not None
        _require_frozen(self)
        ls = resolve_lora_scale(self, scale)
        attn.prepare_attention_mask(attention_mask, 0, 0)
        shape4 = hidden_states.shape if (hidden_states.ndim == 4 and not temporal) else None
        x = hidden_states if temporal else _tok(hidden_states)
        if self.query_condition and self.key_value_condition:
            assert encoder_hidden_states is None
        ctx = x if encoder_hidden_states is None else _tok(encoder_hidden_states)
        pose = _pose_tokens(pose_feature, x)
        fold = self._qkv_fold(attn, x, pose, self.scale, self, ls) if (self.query_condition and self.key_value_condition) else None
        if fold is not None:
            out = _attention_core(attn, x, None, temporal, lora=self, lora_scale=ls, residual=_fusable(attn, _residual, shape4), qkv_fold=fold)
            return _finish(attn, out, hidden_states, shape4)
        q_in, kv_in = self._merge(x, ctx, pose, self.scale)
        out = _attention_core(attn, q_in, kv_in, temporal, lora=self, lora_scale=ls,
                              residual=_fusable(attn, _residual, shape4))
        return _finish(attn, out, hidden_states, shape4)
