"""Tensor-level entry points of the hand-written gfx950 kernels (through the C ABI).

PyTorch is plumbing here: it owns device memory and the stream; every function below hands raw
device pointers + sizes + the current `hipStream_t` to `libfmc_hip.so`.  No CPU / eager fallback:
CPU tensors raise.

Layout: activations are channels-last.  `[N, S, C]` "tokens" are what the kernels see; a
`b c f h w` video in `torch.channels_last_3d` memory format *is* `[(b f), (h w), c]` tokens.
"""
from __future__ import annotations

import atexit
import math
import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import FMC_BF16, FMC_F32

_DT = {torch.bfloat16: FMC_BF16, torch.float32: FMC_F32}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"fmc kernels take bf16 or fp32 activations, got {t.dtype}") from None


def _dev(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("fmc HIP kernels need device (cuda/hip) tensors; there is no CPU fallback")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_ws_cache = {}
_ws_retired = []


def _workspace(device, nbytes: int) -> torch.Tensor:
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        if ws is not None:
            _ws_retired.append(ws)      # a captured HIP graph may have baked its address: never hand it back
        ws = torch.empty(max(nbytes // 4 + 1, 1 << 16), dtype=torch.float32, device=device)
        _ws_cache[key] = ws
    return ws


# --------------------------------------------------------------------------------------------
# GroupNorm (+SiLU)
# --------------------------------------------------------------------------------------------
def groupnorm_silu_raw(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float,
                       act: bool, x2: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """x `[N, S, C]` contiguous -> (y, stats `[N, G, 2]`).  With `x2 [N, S, C2]` the normalised tensor is the channel
    concat `[x, x2]` (never materialised); y is `[N, S, C + C2]`."""
    _dev(x, gamma, beta, x2)
    assert x.ndim == 3 and x.is_contiguous(), "groupnorm: x must be contiguous [N, S, C] tokens"
    N, S, C1 = x.shape
    C = C1
    if x2 is not None:
        assert x2.is_contiguous() and x2.shape[:2] == x.shape[:2] and x2.dtype == x.dtype
        C = C1 + x2.shape[2]
    lib = _lib.load()
    y = torch.empty(N, S, C, dtype=x.dtype, device=x.device)
    stats = torch.empty(N, groups, 2, dtype=torch.float32, device=x.device)
    ws = _workspace(x.device, lib.fmc_groupnorm_workspace_bytes(N, C, groups))
    _lib.check(lib.fmc_groupnorm_silu_fwd(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                          stats.data_ptr(), ws.data_ptr(), N, S, C, groups, float(eps), int(act),
                                          _dt(x), _p(x2), C1 if x2 is not None else 0, _stream()),
               "fmc_groupnorm_silu_fwd")
    return y, stats


# GroupNorm statistics out of the producing GEMM / conv epilogue (fmc_linear_bf16_gn / fmc_conv3x3_bf16_gn -> fmc_groupnorm_apply_fwd).
# A producer that emitted them tags its output tensor: `out._fmc_gn = (partials [n_img, hw / 160, 32, 2], C)`; views made by the layout helpers
# carry the tag, any arithmetic on the tensor makes a new, untagged one.  FMC_GN_EPILOGUE=0: A/B switch.
GN_EPILOGUE = os.environ.get("FMC_GN_EPILOGUE", "1") != "0"
gn_epilogue_calls = {"emitted": 0, "consumed": 0}
GN_MIN_HW = 2048                    # below that the single-pass GroupNorm kernel reads x once anyway


def gn_emit_ok(M: int, N: int, Kd: int, hw: int, dtype) -> bool:
    return (GN_EPILOGUE and dtype == torch.bfloat16 and hw >= GN_MIN_HW and hw % 160 == 0 and M % hw == 0 and N % 320 == 0 and Kd % 64 == 0
            and not torch.is_grad_enabled())


def carry_gn(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    tag = getattr(src, "_fmc_gn", None)
    if tag is not None:
        dst._fmc_gn = tag
    return dst


def groupnorm_apply(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, act: bool,
                    partials: torch.Tensor) -> torch.Tensor:
    """GroupNorm(+SiLU) of `[N, S, C]` tokens whose per-(image, tile, group) sums the producer already wrote (`partials [N, splits, G, 2]`)."""
    _dev(x, gamma, beta, partials)
    N, S, C = x.shape
    assert x.is_contiguous() and partials.shape[0] == N and partials.shape[2] == groups and partials.dtype == torch.float32
    y = torch.empty_like(x)
    stats = torch.empty(N, groups, 2, dtype=torch.float32, device=x.device)
    gn_epilogue_calls["consumed"] += 1
    _lib.check(_lib.load().fmc_groupnorm_apply_fwd(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), stats.data_ptr(),
                                                   partials.data_ptr(), partials.shape[1], N, S, C, groups, float(eps), int(act), _dt(x),
                                                   _stream()), "fmc_groupnorm_apply_fwd")
    return y


class _GroupNormSiLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, act):
        y, stats = groupnorm_silu_raw(x, gamma, beta, groups, eps, act)
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.groups, ctx.act = groups, act
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise NotImplementedError("fmc_groupnorm_silu_bwd computes dX only (gamma/beta are frozen on the FMC path)")
        dy = dy.contiguous()
        N, S, C = x.shape
        lib = _lib.load()
        dx = torch.empty_like(x)
        ws = _workspace(x.device, lib.fmc_groupnorm_workspace_bytes(N, C, ctx.groups))
        _lib.check(lib.fmc_groupnorm_silu_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), gamma.data_ptr(),
                                              beta.data_ptr(), stats.data_ptr(), ws.data_ptr(), N, S, C, ctx.groups,
                                              int(ctx.act), _dt(x), _stream()), "fmc_groupnorm_silu_bwd")
        return dx, None, None, None, None, None


def _gn_backward(ctx, dy, addend):
    x, gamma, beta, stats = ctx.saved_tensors
    if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
        raise NotImplementedError("fmc_groupnorm_silu_bwd computes dX only (gamma/beta are frozen on the FMC path)")
    dy = dy.contiguous()
    N, S, C = x.shape
    lib = _lib.load()
    dx = torch.empty_like(x)
    ws = _workspace(x.device, lib.fmc_groupnorm_workspace_bytes(N, C, ctx.groups))
    _lib.check(lib.fmc_groupnorm_silu_bwd_add(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), gamma.data_ptr(), beta.data_ptr(), stats.data_ptr(),
                                              ws.data_ptr(), N, S, C, ctx.groups, int(ctx.act), _p(addend), _dt(x), _stream()),
               "fmc_groupnorm_silu_bwd_add")
    return dx


class _GroupNormSiLUSkip(torch.autograd.Function):
    """`(h, GroupNorm(h))` as ONE autograd node for `h + f(norm(h))`: the backward receives the gradient along the skip and the one through
    the norm together and returns `d_skip + dX(norm)` from the norm's backward kernel (`fmc_groupnorm_silu_bwd_add`) -- autograd would
    otherwise sum the two with an elementwise launch per residual connection (303 per OMC-stage step)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, act):
        y, stats = groupnorm_silu_raw(x, gamma, beta, groups, eps, act)
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.groups, ctx.act = groups, act
        ctx.set_materialize_grads(False)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, d_skip, dy):
        if dy is None:
            return d_skip, None, None, None, None, None
        add = None
        if d_skip is not None:
            add = d_skip if d_skip.is_contiguous() else d_skip.contiguous()
        return _gn_backward(ctx, dy, add), None, None, None, None, None


def groupnorm_silu_skip(x, gamma, beta, groups: int, eps: float, act: bool):
    """(x for the skip connection, GroupNorm(+SiLU)(x)) -- one autograd node when x needs a gradient, the plain pair otherwise."""
    if torch.is_grad_enabled() and x.requires_grad and x.is_cuda and x.is_contiguous():
        return _GroupNormSiLUSkip.apply(x, gamma, beta, groups, eps, act)
    return x, groupnorm_silu(x, gamma, beta, groups, eps, act)


def groupnorm_silu(x, gamma, beta, groups: int, eps: float, act: bool, x2=None, gn_tag=None) -> torch.Tensor:
    """GroupNorm over `[N, S, C]` tokens (statistics per sample and group over S x C/G) + optional SiLU.
    gamma/beta: fp32 `[C]`.  `x2`: second channel block (the result normalises `cat([x, x2], -1)` without building it).
    `gn_tag`: the `_fmc_gn` tag of the tensor x is a view of -- its producer's partial sums replace the statistics pass."""
    if (gn_tag is not None and x2 is None and groups == 32 and gn_tag[1] == x.shape[-1] and gn_tag[0].shape[0] == x.shape[0]
            and gn_tag[0].shape[1] <= 64 and not torch.is_grad_enabled()):       # (any split of the pixels: 160-row tiles, the halo conv's 10 x 32 tiles)
        return groupnorm_apply(x, gamma, beta, groups, eps, act, gn_tag[0])
    if torch.is_grad_enabled() and (x.requires_grad or (x2 is not None and x2.requires_grad)):
        if x2 is not None:
            x = torch.cat([x, x2], dim=-1)
        return _GroupNormSiLU.apply(x, gamma, beta, groups, eps, act)
    return groupnorm_silu_raw(x, gamma, beta, groups, eps, act, x2)[0]


# --------------------------------------------------------------------------------------------
# LayerNorm (+ positional encoding), GEGLU
# --------------------------------------------------------------------------------------------
class _LayerNorm(torch.autograd.Function):
    """gamma / beta here are the fp32 views the kernel reads; their grads come back in fp32."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, pe, pe_inner, pe_frames):
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        return _layernorm_raw(x, gamma, beta, eps, pe, pe_inner, pe_frames)

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dy = dy.contiguous()
        C = x.shape[-1]
        M = x.numel() // C
        dx = torch.empty_like(x)
        want_p = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dg = torch.zeros(C, dtype=torch.float32, device=x.device) if want_p else None
        db = torch.zeros(C, dtype=torch.float32, device=x.device) if want_p else None
        _lib.check(_lib.load().fmc_layernorm_bwd(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), dx.data_ptr(), _p(dg),
                                                 _p(db), M, C, float(ctx.eps), _dt(x), _stream()), "fmc_layernorm_bwd")
        return dx, dg, db, None, None, None, None


def _ln_backward(ctx, dy, addend):
    x, gamma = ctx.saved_tensors
    dy = dy.contiguous()
    C = x.shape[-1]
    M = x.numel() // C
    dx = torch.empty_like(x)
    want_p = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
    dg = torch.zeros(C, dtype=torch.float32, device=x.device) if want_p else None
    db = torch.zeros(C, dtype=torch.float32, device=x.device) if want_p else None
    _lib.check(_lib.load().fmc_layernorm_bwd_add(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), dx.data_ptr(), _p(dg), _p(db), _p(addend), M, C,
                                                 float(ctx.eps), _dt(x), _stream()), "fmc_layernorm_bwd_add")
    return dx, dg, db


class _LayerNormSkip(torch.autograd.Function):
    """`(h, LayerNorm(h) [+ pe])` as ONE autograd node (see _GroupNormSiLUSkip): backward = `d_skip + dX(norm)` in `fmc_layernorm_bwd_add`."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, pe, pe_inner, pe_frames):
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        ctx.set_materialize_grads(False)
        return x.view_as(x), _layernorm_raw(x, gamma, beta, eps, pe, pe_inner, pe_frames)

    @staticmethod
    def backward(ctx, d_skip, dy):
        if dy is None:
            return d_skip, None, None, None, None, None, None
        add = None
        if d_skip is not None:
            add = d_skip if d_skip.is_contiguous() else d_skip.contiguous()
        dx, dg, db = _ln_backward(ctx, dy, add)
        return dx, dg, db, None, None, None, None


def layernorm_skip(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
                   pe: Optional[torch.Tensor] = None, pe_inner: int = 1, pe_frames: int = 1):
    """(x for the skip connection, LayerNorm(x)) -- one autograd node when x needs a gradient."""
    if torch.is_grad_enabled() and x.requires_grad and x.is_cuda and x.is_contiguous():
        return _LayerNormSkip.apply(x, gamma, beta, eps, pe, pe_inner, pe_frames)
    return x, layernorm(x, gamma, beta, eps, pe, pe_inner, pe_frames)


LAZY_RESIDUAL = os.environ.get("FMC_LAZY_RESIDUAL", "1") != "0"    # A/B switch: the residual add behind a vendor-arm projection rides in the next LayerNorm


def layernorm_add(x: torch.Tensor, addend: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
                  pe: Optional[torch.Tensor] = None, pe_inner: int = 1, pe_frames: int = 1):
    """`h = x + addend` (rounded to the storage type, as a separate add would) and `LayerNorm(h) (+ pe)` in one pass: returns (h, y)."""
    _dev(x, addend, gamma, beta, pe)
    assert x.is_contiguous() and addend.is_contiguous() and x.shape == addend.shape and x.dtype == addend.dtype
    C = x.shape[-1]
    M = x.numel() // C
    h, y = torch.empty_like(x), torch.empty_like(x)
    _lib.check(_lib.load().fmc_layernorm_add_fwd(x.data_ptr(), addend.data_ptr(), h.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                 _p(pe), M, C, float(eps), int(pe_inner), int(pe_frames), _dt(x), _stream()), "fmc_layernorm_add_fwd")
    return h, y


def resolve_pending_add(x: torch.Tensor) -> torch.Tensor:
    """A projection output whose residual add was left to its consumer (`linear(..., lazy_residual=True)` on the vendor arm): the sum."""
    r = getattr(x, "_fmc_pending_add", None)
    return x if r is None else torch.add(r, x)


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
              pe: Optional[torch.Tensor] = None, pe_inner: int = 1, pe_frames: int = 1) -> torch.Tensor:
    """LayerNorm over the last dim of contiguous tokens; optionally adds `pe[(row // pe_inner) % pe_frames]`
    (fp32 `[>=pe_frames, C]`) after normalising."""
    if torch.is_grad_enabled() and (x.requires_grad or gamma.requires_grad or beta.requires_grad):
        return _LayerNorm.apply(x, gamma, beta, eps, pe, pe_inner, pe_frames)
    return _layernorm_raw(x, gamma, beta, eps, pe, pe_inner, pe_frames)


def _layernorm_raw(x, gamma, beta, eps, pe, pe_inner, pe_frames):
    _dev(x, gamma, beta, pe)
    assert x.is_contiguous()
    C = x.shape[-1]
    M = x.numel() // C
    y = torch.empty_like(x)
    _lib.check(_lib.load().fmc_layernorm_fwd(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _p(pe), M,
                                             C, float(eps), int(pe_inner), int(pe_frames), _dt(x), _stream()),
               "fmc_layernorm_fwd")
    return y


class _Geglu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return _geglu_raw(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        cff = x.shape[-1] // 2
        dx = torch.empty_like(x)
        _lib.check(_lib.load().fmc_geglu_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel() // (2 * cff), cff,
                                             _dt(x), _stream()), "fmc_geglu_bwd")
        return dx


def geglu(x: torch.Tensor) -> torch.Tensor:
    """`a * gelu_erf(g)` with `a, g = x.chunk(2, -1)`; x contiguous `[..., 2*Cff]`."""
    if torch.is_grad_enabled() and x.requires_grad:
        return _Geglu.apply(x)
    return _geglu_raw(x)


def _geglu_raw(x: torch.Tensor) -> torch.Tensor:
    _dev(x)
    assert x.is_contiguous()
    cff = x.shape[-1] // 2
    M = x.numel() // (2 * cff)
    y = torch.empty(*x.shape[:-1], cff, dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().fmc_geglu_fwd(x.data_ptr(), y.data_ptr(), M, cff, _dt(x), _stream()), "fmc_geglu_fwd")
    return y


# --------------------------------------------------------------------------------------------
# attention
# --------------------------------------------------------------------------------------------
def _rows(t: torch.Tensor) -> Tuple[int, int]:
    """(batch stride, row stride) in elements of a `[B, S, C]` view whose last dim is dense."""
    assert t.ndim == 3 and t.stride(2) == 1, "attention operands must be [B, S, C] with a dense channel dim"
    return t.stride(0), t.stride(1)


class _SpatialAttention(torch.autograd.Function):
    """q `[B,S,C]`, kv-side tensors `[Bkv,Skv,C]` (strided slices allowed).  Grads come back as dense tensors."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale):
        o, lse = _spatial_attention_raw(q, k, v, heads, scale, True)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.heads, ctx.scale = heads, scale
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, o, lse = ctx.saved_tensors
        B, Sq, C = q.shape
        Bkv, Skv, _ = k.shape
        heads = ctx.heads
        D = C // heads
        d_o = d_o.contiguous()
        dq = torch.empty(B, Sq, C, dtype=q.dtype, device=q.device)
        need_kv = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dk = torch.empty(Bkv, Skv, C, dtype=q.dtype, device=q.device) if need_kv else None
        dv = torch.empty(Bkv, Skv, C, dtype=q.dtype, device=q.device) if need_kv else None
        dvec = torch.empty(B, heads, Sq, dtype=torch.float32, device=q.device)
        _lib.check(_lib.load().fmc_spatial_attn_bwd(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), dvec.data_ptr(),
            dq.data_ptr(), _p(dk), _p(dv), B, heads, Sq, Skv, D, q.stride(0), q.stride(1), k.stride(0),
            k.stride(1), Sq * C, C, Sq * C, C, Skv * C, C, B // Bkv, float(ctx.scale), _dt(q), _stream()),
            "fmc_spatial_attn_bwd")
        return dq, dk, dv, None, None


def spatial_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: Optional[float] = None,
                      return_lse: bool = False):
    """softmax(Q K^T * scale) V per (batch, head); differentiable (fmc_spatial_attn_bwd)."""
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad) and not return_lse:
        sc = (q.shape[-1] // heads) ** -0.5 if scale is None else scale
        return _SpatialAttention.apply(q, k, v, heads, sc)
    return _spatial_attention_raw(q, k, v, heads, scale, return_lse)


def _spatial_attention_raw(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: Optional[float] = None,
                           return_lse: bool = False):
    """softmax(Q K^T * scale) V per (batch, head).  q `[B, Sq, H*D]`, k/v `[Bkv, Skv, H*D]` (strided views of a
    fused projection are fine; B must be a multiple of Bkv: batch b reads kv batch b // (B // Bkv))."""
    _dev(q, k, v)
    B, Sq, C = q.shape
    Bkv, Skv, _ = k.shape
    assert C % heads == 0 and k.shape == v.shape and k.shape[2] == C and B % Bkv == 0
    assert k.stride() == v.stride(), "k and v must share strides (slices of one fused projection)"
    D = C // heads
    scale = D ** -0.5 if scale is None else scale
    o = torch.empty(B, Sq, C, dtype=q.dtype, device=q.device)
    lse = torch.empty(B, heads, Sq, dtype=torch.float32, device=q.device) if return_lse else None
    qb, qr = _rows(q)
    kb, kr = _rows(k)
    _lib.check(_lib.load().fmc_spatial_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), _p(lse), B,
                                                heads, Sq, Skv, D, qb, qr, kb, kr, Sq * C, C, B // Bkv, float(scale),
                                                _dt(q), _stream()), "fmc_spatial_attn_fwd")
    return (o, lse) if return_lse else o


def attention_ok(head_dim: int) -> bool:
    return head_dim in (32, 64, 128, 256, 512)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: Optional[float] = None, causal: bool = False,
              key_keep: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(Q K^T scale + mask) V per (batch, head) on `fmc_attention_fwd` (csrc/attn_generic.hip): head widths 32 .. 512 and masked scores -- the VAE
    mid block's single head of width 512 and CLIP's causal text attention (once per clip, no autograd).  q `[B, Sq, H D]`, k / v
    `[B, Skv, H D]` (strided views of a fused projection are fine), `key_keep [B, Skv]` bool / uint8 (True = attend) or None."""
    _dev(q, k, v, key_keep)
    B, Sq, C = q.shape
    Bk, Skv, _ = k.shape
    assert C % heads == 0 and k.shape == v.shape and k.shape[2] == C and Bk == B and k.stride() == v.stride() and q.dtype == k.dtype == v.dtype
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        raise NotImplementedError("hip_ops.attention is forward-only (VAE / text encoder are frozen in every FMC stage)")
    D = C // heads
    if not _lib.load().fmc_attention_supported(D):
        raise ValueError(f"hip_ops.attention: head width {D} (supported: 32, 64, 128, 256, 512)")
    scale = D ** -0.5 if scale is None else scale
    o = torch.empty(B, Sq, C, dtype=q.dtype, device=q.device)
    keep = None
    if key_keep is not None:
        keep = key_keep.to(torch.uint8).contiguous()
        assert tuple(keep.shape) == (B, Skv)
    qb, qr = _rows(q)
    kb, kr = _rows(k)
    _lib.check(_lib.load().fmc_attention_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), _p(keep), B, heads, Sq, Skv, D, qb, qr, kb, kr,
                                             Sq * C, C, float(scale), int(causal), _dt(q), _stream()), "fmc_attention_fwd")
    return o


def _tstrides(t):
    if t.ndim == 4:
        return t.shape[0], t.shape[2], t.shape[1], t.stride(0), t.stride(1), t.stride(2)
    return 1, t.shape[0], t.shape[1], 0, t.stride(1), t.stride(0)


class _TemporalAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, heads, scale):
        ctx.save_for_backward(q, k, v)
        ctx.heads, ctx.scale = heads, scale
        return _temporal_attention_raw(q, k, v, heads, scale)

    @staticmethod
    def backward(ctx, d_o):
        q, k, v = ctx.saved_tensors
        d_o = d_o.contiguous()
        C = q.shape[-1]
        dqkv = torch.empty(*q.shape[:-1], 3 * C, dtype=q.dtype, device=q.device)   # fused [.., 3C] gradient
        dq, dk, dv = dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:]
        B, P, F, cs, fs, ps = _tstrides(q)
        _, _, _, ocs, ofs, ops = _tstrides(d_o)
        _, _, _, dcs, dfs, dps = _tstrides(dq)
        _lib.check(_lib.load().fmc_temporal_attn_bwd(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), d_o.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, P,
            F, ctx.heads, C // ctx.heads, cs, fs, ps, ocs, ofs, ops, dcs, dfs, dps, float(ctx.scale), _dt(q), _stream()),
            "fmc_temporal_attn_bwd")
        return dq, dk, dv, None, None


class _SelfAttentionQKV(torch.autograd.Function):
    """Self attention on the fused projection output `qkv [.., 3C]`; the backward kernels write dQ | dK | dV straight
    into ONE `[.., 3C]` gradient through their stride arguments.  (Passing the three slices through autograd instead
    costs, per attention, three zero-fills of the fused shape, three slice copies and two adds.)"""

    @staticmethod
    def forward(ctx, qkv, heads, scale, temporal):
        C = qkv.shape[-1] // 3
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        ctx.heads, ctx.scale, ctx.temporal = heads, scale, temporal
        if temporal:
            ctx.save_for_backward(qkv)
            return _temporal_attention_raw(q, k, v, heads, scale)
        o, lse = _spatial_attention_raw(q, k, v, heads, scale, True)
        ctx.save_for_backward(qkv, o, lse)
        return o

    @staticmethod
    def backward(ctx, d_o):
        qkv = ctx.saved_tensors[0]
        C = qkv.shape[-1] // 3
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        d_o = d_o.contiguous()
        dqkv = torch.empty(qkv.shape, dtype=qkv.dtype, device=qkv.device)
        dq, dk, dv = dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:]
        heads = ctx.heads
        if ctx.temporal:
            B, P, F, cs, fs, ps = _tstrides(q)
            _, _, _, ocs, ofs, ops = _tstrides(d_o)
            _, _, _, dcs, dfs, dps = _tstrides(dq)
            _lib.check(_lib.load().fmc_temporal_attn_bwd(
                q.data_ptr(), k.data_ptr(), v.data_ptr(), d_o.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B,
                P, F, heads, C // heads, cs, fs, ps, ocs, ofs, ops, dcs, dfs, dps, float(ctx.scale), _dt(q), _stream()),
                "fmc_temporal_attn_bwd")
        else:
            _, o, lse = ctx.saved_tensors
            B, S, _ = q.shape
            dvec = torch.empty(B, heads, S, dtype=torch.float32, device=q.device)
            _lib.check(_lib.load().fmc_spatial_attn_bwd(
                q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), dvec.data_ptr(),
                dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, heads, S, S, C // heads, q.stride(0), q.stride(1),
                k.stride(0), k.stride(1), S * C, C, dq.stride(0), dq.stride(1), dk.stride(0), dk.stride(1), 1,
                float(ctx.scale), _dt(q), _stream()), "fmc_spatial_attn_bwd")
        return dqkv, None, None, None


class _CrossAttentionQKV(torch.autograd.Function):
    """Cross attention with the fused `kv [Bkv, Skv, 2C]` projection; one fused dK | dV gradient."""

    @staticmethod
    def forward(ctx, q, kv, heads, scale):
        C = q.shape[-1]
        o, lse = _spatial_attention_raw(q, kv[..., :C], kv[..., C:], heads, scale, True)
        ctx.save_for_backward(q, kv, o, lse)
        ctx.heads, ctx.scale = heads, scale
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, kv, o, lse = ctx.saved_tensors
        B, Sq, C = q.shape
        Bkv, Skv, _ = kv.shape
        k, v = kv[..., :C], kv[..., C:]
        heads = ctx.heads
        d_o = d_o.contiguous()
        dq = torch.empty(B, Sq, C, dtype=q.dtype, device=q.device)
        dvec = torch.empty(B, heads, Sq, dtype=torch.float32, device=q.device)
        div, kbs = B // Bkv, k.stride(0)
        if not ctx.needs_input_grad[1]:
            # frozen to_k / to_v on a constant text embedding (every FMC training stage): no dK / dV kernel at all
            dkv = dk = dv = None
        elif Bkv == 1 and div > 1:
            # one text for all frames: per-frame partial dK | dV (batch stride 0 on K / V keeps B x heads workgroups
            # busy instead of `heads`), summed over the frames afterwards
            part = torch.empty(B, Skv, 2 * C, dtype=q.dtype, device=q.device)
            dk, dv, div, kbs, dkv = part[..., :C], part[..., C:], 1, 0, part
        else:
            dkv = torch.empty(kv.shape, dtype=q.dtype, device=q.device)
            dk, dv = dkv[..., :C], dkv[..., C:]
        _lib.check(_lib.load().fmc_spatial_attn_bwd(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), dvec.data_ptr(),
            dq.data_ptr(), _p(dk), _p(dv), B, heads, Sq, Skv, C // heads, q.stride(0), q.stride(1), kbs, k.stride(1),
            Sq * C, C, Sq * C, C, dk.stride(0) if dk is not None else 0, dk.stride(1) if dk is not None else 2 * C, div,
            float(ctx.scale), _dt(q), _stream()), "fmc_spatial_attn_bwd")
        if dkv is not None and dkv.shape[0] != Bkv:
            dkv = dkv.sum(0, keepdim=True, dtype=torch.float32).to(q.dtype)
        return dq, dkv, None, None


def self_attention_qkv(qkv: torch.Tensor, heads: int, scale: float, temporal: bool) -> torch.Tensor:
    """Attention on a fused `[.., 3C]` projection (q | k | v); differentiable with a single fused gradient."""
    C = qkv.shape[-1] // 3
    if torch.is_grad_enabled() and qkv.requires_grad:
        return _SelfAttentionQKV.apply(qkv, heads, scale, temporal)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    return _temporal_attention_raw(q, k, v, heads, scale) if temporal else _spatial_attention_raw(q, k, v, heads, scale)


def cross_attention_q_kv(q: torch.Tensor, kv: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    """Cross attention with a fused `[.., 2C]` (k | v) projection; differentiable with a fused dK | dV gradient."""
    C = q.shape[-1]
    if torch.is_grad_enabled() and (q.requires_grad or kv.requires_grad):
        return _CrossAttentionQKV.apply(q, kv, heads, scale)
    return _spatial_attention_raw(q, kv[..., :C], kv[..., C:], heads, scale)


def temporal_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int,
                       scale: Optional[float] = None) -> torch.Tensor:
    """Attention over the frame axis (see `_temporal_attention_raw`); differentiable (fmc_temporal_attn_bwd)."""
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        sc = (q.shape[-1] // heads) ** -0.5 if scale is None else scale
        return _TemporalAttention.apply(q, k, v, heads, sc)
    return _temporal_attention_raw(q, k, v, heads, scale)


def _temporal_attention_raw(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int,
                            scale: Optional[float] = None) -> torch.Tensor:
    """Attention over the frame axis.  Accepts the native layout `[B, F, P, C]` (frames outer, pixels inner: the
    channels-last video) or the reference layout `[N, F, C]` (`(b h w) f c`).  q/k/v may be strided slices of a
    fused QKV tensor (same strides for all three).  Returns a contiguous tensor of q's shape."""
    _dev(q, k, v)
    assert q.shape == k.shape == v.shape and q.stride() == k.stride() == v.stride() and q.stride(-1) == 1
    C = q.shape[-1]
    D = C // heads
    scale = D ** -0.5 if scale is None else scale
    o = torch.empty(q.shape, dtype=q.dtype, device=q.device)
    if q.ndim == 4:
        B, F, P, _ = q.shape
        cs, fs, ps = q.stride(0), q.stride(1), q.stride(2)
        ocs, ofs, ops = o.stride(0), o.stride(1), o.stride(2)
    elif q.ndim == 3:
        P, F, _ = q.shape
        B = 1
        cs, fs, ps = 0, q.stride(1), q.stride(0)
        ocs, ofs, ops = 0, o.stride(1), o.stride(0)
    else:
        raise ValueError("temporal_attention expects [B, F, P, C] or [N, F, C]")
    _lib.check(_lib.load().fmc_temporal_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), B, P, F, heads,
                                                 D, cs, fs, ps, ocs, ofs, ops, float(scale), _dt(q), _stream()),
               "fmc_temporal_attn_fwd")
    return o


# --------------------------------------------------------------------------------------------
# fp8 (e4m3) temporal attention: QKV projection with an fp8 epilogue + attention on fp8 MFMA (BASELINE configs[4])
# --------------------------------------------------------------------------------------------
FP8_MAX = 448.0


class Fp8QKVScales:
    """Per-attention-module scale state for the fp8 q | k | v tensors (delayed scaling: the scales of call t come from the
    running |max| the projection's epilogue recorded in call t - 1).  Everything lives on the device and is updated by
    device ops only, so a captured HIP graph replays it.  `margin` leaves headroom for a max that grows between calls
    (values beyond it saturate at +-448 * scale)."""

    def __init__(self, device, margin: float = 1.25):
        self.margin = margin
        self.amax = torch.zeros(3, dtype=torch.float32, device=device)       # float bit patterns, written by atomicMax
        self.scale = torch.ones(3, dtype=torch.float32, device=device)
        self.inv_scale = torch.ones(3, dtype=torch.float32, device=device)
        self.calibrated = False

    def calibrate(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> None:
        """First call: take the maxima from a bf16 projection."""
        self.amax.copy_(torch.stack([q.abs().amax(), k.abs().amax(), v.abs().amax()]).float())
        self.calibrated = True

    def roll(self) -> None:
        """scale <- margin * amax / 448, inv_scale <- 1 / scale, then restart the running max: one tiny launch."""
        _lib.check(_lib.load().fmc_fp8_scales_roll(self.amax.data_ptr(), self.scale.data_ptr(), self.inv_scale.data_ptr(),
                                                   float(self.margin), _stream()), "fmc_fp8_scales_roll")


def linear_fp8_qkv(x: torch.Tensor, weight: torch.Tensor, scales: Fp8QKVScales) -> torch.Tensor:
    """`x @ weight^T` -> `[..., 3C]` e4m3 bytes (torch.float8_e4m3fn), block b scaled by `scales.inv_scale[b]`; records |max|."""
    _dev(x, weight)
    N, Kd = weight.shape
    assert x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and weight.is_contiguous()
    assert N % 192 == 0 and Kd % 64 == 0, "fp8 QKV projection needs C % 64 == 0 and K % 64 == 0"
    M, ldx = _rows2d(x)
    out = torch.empty(*x.shape[:-1], N, dtype=torch.float8_e4m3fn, device=x.device)
    _lib.check(_lib.load().fmc_linear_fp8_qkv(x.data_ptr(), weight.data_ptr(), out.data_ptr(), M, N, Kd, ldx,
                                              scales.inv_scale.data_ptr(), scales.amax.data_ptr(), _stream()),
               "fmc_linear_fp8_qkv")
    return out


def _temporal_fp8_raw(qkv8: torch.Tensor, scale_dev: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    C = qkv8.shape[-1] // 3
    q, k, v = qkv8[..., :C], qkv8[..., C:2 * C], qkv8[..., 2 * C:]
    o = torch.empty(*qkv8.shape[:-1], C, dtype=torch.bfloat16, device=qkv8.device)
    B, P, F, cs, fs, ps = _tstrides(q)
    _, _, _, ocs, ofs, ops = _tstrides(o)
    _lib.check(_lib.load().fmc_temporal_attn_fp8_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(),
                                                     scale_dev.data_ptr(), B, P, F, heads, C // heads, cs, fs, ps, ocs, ofs,
                                                     ops, float(scale), _stream()), "fmc_temporal_attn_fp8_fwd")
    return o


class _TemporalAttentionFp8(torch.autograd.Function):
    """Projection (fp8 epilogue) + fp8 temporal attention as ONE autograd node: x -> o.  Backward: the attention backward
    kernel dequantises the saved e4m3 q | k | v while staging them (straight-through w.r.t. the quantisation), writes one
    fused bf16 dQ | dK | dV, and the projection's input gradient is `dqkv @ W` (frozen weight) -- or dW as well when the
    weight trains (camera encoder)."""

    @staticmethod
    def forward(ctx, x, weight, scales, heads, scale):
        qkv8 = linear_fp8_qkv(x, weight, scales)
        sc = scales.scale.clone()                     # the scales THIS call used (the state rolls on)
        ctx.save_for_backward(x, weight, qkv8, sc)
        ctx.heads, ctx.scale = heads, scale
        return _temporal_fp8_raw(qkv8, sc, heads, scale)

    @staticmethod
    def backward(ctx, d_o):
        x, weight, qkv8, sc = ctx.saved_tensors
        C = qkv8.shape[-1] // 3
        q, k, v = qkv8[..., :C], qkv8[..., C:2 * C], qkv8[..., 2 * C:]
        d_o = d_o.contiguous()
        dqkv = torch.empty(qkv8.shape, dtype=torch.bfloat16, device=qkv8.device)
        dq, dk, dv = dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:]
        B, P, F, cs, fs, ps = _tstrides(q)
        _, _, _, ocs, ofs, ops = _tstrides(d_o)
        _, _, _, dcs, dfs, dps = _tstrides(dq)
        _lib.check(_lib.load().fmc_temporal_attn_fp8_bwd(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), sc.data_ptr(), d_o.data_ptr(), dq.data_ptr(), dk.data_ptr(),
            dv.data_ptr(), B, P, F, ctx.heads, C // ctx.heads, cs, fs, ps, ocs, ofs, ops, dcs, dfs, dps, float(ctx.scale),
            _stream()), "fmc_temporal_attn_fp8_bwd")
        dx = None
        if ctx.needs_input_grad[0]:                  # frozen weight: own GEMM on the cached W^T; a training weight: plain matmul
            dx = torch.matmul(dqkv, weight) if weight.requires_grad else linear_backward_data(dqkv, weight)
        dw = None
        if ctx.needs_input_grad[1]:
            dw = torch.matmul(dqkv.reshape(-1, dqkv.shape[-1]).t(), x.reshape(-1, x.shape[-1]))
        return dx, dw, None, None, None


def temporal_attention_fp8(x: torch.Tensor, weight: torch.Tensor, scales: Fp8QKVScales, heads: int, scale: float):
    """o = temporal_attention(split(x @ W_qkv^T)) with e4m3 q | k | v (native `[B, F, P, C]` tokens or `[N, F, C]`).
    The first call on a fresh `scales` runs the bf16 path once to calibrate the maxima."""
    if not scales.calibrated:
        with torch.no_grad():
            qkv = linear(x.detach(), weight.detach())
            C = qkv.shape[-1] // 3
            scales.calibrate(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:])
    scales.roll()
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        return _TemporalAttentionFp8.apply(x, weight, scales, heads, scale)
    qkv8 = linear_fp8_qkv(x, weight, scales)
    return _temporal_fp8_raw(qkv8, scales.scale, heads, scale)


# --------------------------------------------------------------------------------------------
# conditioning
# --------------------------------------------------------------------------------------------
def plucker(K: torch.Tensor, c2w: torch.Tensor, H: int, W: int, layout: str = "bfhwc",
            dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """Pluecker embedding on device.  K `[B, F, 4]`, c2w `[B, F, 3|4, 4]` (fp32).
    layout "bfhwc": `[B,F,H,W,6]` (= ray_condition); "bcfhw": `[B,6,F,H,W]`;
    "unshuffle8": `[B*F, H/8, W/8, 384]` channels-last with PixelUnshuffle(8) applied."""
    _dev(K, c2w)
    B, F = K.shape[:2]
    K = K.to(torch.float32).contiguous()
    c2w = c2w.to(torch.float32).contiguous()
    rows = c2w.shape[2]
    code = {"bfhwc": 0, "bcfhw": 1, "unshuffle8": 2}[layout]
    shape = {0: (B, F, H, W, 6), 1: (B, 6, F, H, W), 2: (B * F, H // 8, W // 8, 384)}[code]
    out = torch.empty(shape, dtype=dtype, device=K.device)
    _lib.check(_lib.load().fmc_plucker_fwd(K.data_ptr(), c2w.data_ptr(), out.data_ptr(), B, F, H, W, rows, code,
                                           _dt(out), _stream()), "fmc_plucker_fwd")
    return out


def gaussian_circle_masks(circles: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """circles `[..., 3]` = (cx, cy, radius) pixels -> Gaussian circle masks `[..., H, W]` fp32 (dataset.py:5365-5380)."""
    _dev(circles)
    c = circles.to(torch.float32).contiguous()
    out = torch.empty(*c.shape[:-1], H, W, dtype=torch.float32, device=c.device)
    _lib.check(_lib.load().fmc_gaussian_circle_mask_fwd(c.data_ptr(), out.data_ptr(), c.numel() // 3, H, W, _stream()),
               "fmc_gaussian_circle_mask_fwd")
    return out


def omc_rasterize(poses: torch.Tensor, masks: torch.Tensor, layout: str = "planar",
                  dtype: torch.dtype = torch.float32) -> Tuple[torch.Tensor, torch.Tensor]:
    """poses `[BF, n_obj, 12]`, masks `[BF, n_obj, H, W]` (fp32) -> (features, mask).
    "planar": features `[BF,13,H,W]`, mask `[BF,1,H,W]`; "unshuffle8": features `[BF,H/8,W/8,832]`, mask `[BF,H,W]`."""
    _dev(poses, masks)
    BF, n_obj, H, W = masks.shape
    poses = poses.to(torch.float32).contiguous()
    masks = masks.to(torch.float32).contiguous()
    code = {"planar": 0, "unshuffle8": 2}[layout]
    feat = torch.empty((BF, 13, H, W) if code == 0 else (BF, H // 8, W // 8, 832), dtype=dtype, device=masks.device)
    mout = torch.empty((BF, 1, H, W) if code == 0 else (BF, H, W), dtype=torch.float32, device=masks.device)
    _lib.check(_lib.load().fmc_omc_rasterize_fwd(poses.data_ptr(), masks.data_ptr(), feat.data_ptr(), mout.data_ptr(),
                                                 BF, n_obj, H, W, code, _dt(feat), _stream()), "fmc_omc_rasterize_fwd")
    return feat, mout


def _mask_modulate_raw(x, mask_in, h, w, want_mask):
    N, S, C = x.shape
    y = torch.empty_like(x)
    mo = torch.empty(N, h, w, dtype=torch.float32, device=x.device) if want_mask else None
    _lib.check(_lib.load().fmc_mask_modulate_fwd(x.data_ptr(), mask_in.data_ptr(), y.data_ptr(), _p(mo), N, h, w, C,
                                                 mask_in.shape[1], mask_in.shape[2], _dt(x), _stream()),
               "fmc_mask_modulate_fwd")
    return y, mo


class _MaskModulate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask_in, h, w):
        y, mo = _mask_modulate_raw(x, mask_in, h, w, True)
        ctx.save_for_backward(mask_in)
        ctx.hw = (h, w)
        ctx.mark_non_differentiable(mo)
        return y, mo

    @staticmethod
    def backward(ctx, dy, _dmo):
        (mask_in,) = ctx.saved_tensors
        dx, _ = _mask_modulate_raw(dy.contiguous(), mask_in, ctx.hw[0], ctx.hw[1], False)
        return dx, None, None, None


def mask_modulate(x: torch.Tensor, mask_in: torch.Tensor, h: int, w: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """`x * nearest(mask_in -> h x w)`; x `[N, h*w, C]` tokens, mask_in `[N, Hin, Win]` fp32.
    Returns (y, resampled mask `[N, h, w]`)."""
    _dev(x, mask_in)
    assert x.is_contiguous() and mask_in.is_contiguous() and mask_in.dtype == torch.float32
    assert x.shape[1] == h * w
    if torch.is_grad_enabled() and x.requires_grad:
        return _MaskModulate.apply(x, mask_in, h, w)
    return _mask_modulate_raw(x, mask_in, h, w, True)


class _FeatureAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, t, skip):
        ctx.skip, ctx.tshape = skip, t.shape
        return _feature_add_raw(h, t, skip, False)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        return dy, dy.reshape(-1)[ctx.skip:].reshape(ctx.tshape), None


def _feature_add_raw(h, t, skip, inplace):
    out = h if inplace else torch.empty_like(h)
    _lib.check(_lib.load().fmc_feature_add_fwd(h.data_ptr(), t.data_ptr(), out.data_ptr(), h.numel(), skip, _dt(h),
                                               _stream()), "fmc_feature_add_fwd")
    return out


def feature_add(h: torch.Tensor, t: torch.Tensor, inplace: bool = False) -> torch.Tensor:
    """`h + t` where `t` covers only the LAST `t.numel()` elements of `h` (the conditioned half under
    classifier-free guidance; the leading elements get nothing added)."""
    _dev(h, t)
    assert h.is_contiguous() and t.is_contiguous() and h.dtype == t.dtype and t.numel() <= h.numel()
    skip = h.numel() - t.numel()
    if torch.is_grad_enabled() and (h.requires_grad or t.requires_grad):
        out = _FeatureAdd.apply(h, t, skip)
        return out
    return _feature_add_raw(h, t, skip, inplace)


def cfg_ddim_step(eps: torch.Tensor, x: torch.Tensor, guidance: float, alpha_t: float, alpha_prev: float,
                  has_uncond: bool) -> torch.Tensor:
    """Fused classifier-free-guidance combine + DDIM(eta=0) update.  eps `[2B, ...]` (uncond || cond) or `[B, ...]`;
    x fp32 latents `[B, ...]`.  Returns the new fp32 latents."""
    _dev(eps, x)
    assert x.dtype == torch.float32 and x.is_contiguous() and eps.is_contiguous()
    n = x.numel()
    assert eps.numel() == (2 * n if has_uncond else n)
    out = torch.empty_like(x)
    _lib.check(_lib.load().fmc_cfg_ddim_step(eps.data_ptr(), x.data_ptr(), out.data_ptr(), n, int(has_uncond),
                                             float(guidance), float(alpha_t), float(alpha_prev), _dt(eps), _stream()),
               "fmc_cfg_ddim_step")
    return out


# --------------------------------------------------------------------------------------------
# bf16 MFMA GEMM / implicit 3x3 conv with fused epilogues
# --------------------------------------------------------------------------------------------
ARM_160 = 512
ARM_256 = 528                       # C-ABI tile 17: persistent 256 x 320 tiles, GEGLU projections only (falls back to tile 16)
ARM_SMALLM = 600                    # C-ABI tiles 19 .. 22: 64 x 128 (4 waves, 64- / 32-deep k-tiles), 128 x 128 and 64 x 256 (8 waves) with 32 x 64 per wave, for M <= 2560 (round 4)
ARM_G4 = 700                        # `fmc_linear4_bf16` (csrc/gemm4.hip, round 5): 160 x 160 tiles, 4 waves, software-pipelined -- the M <= 5120 projections of the inner levels
ARM_160B = 544                      # C-ABI tile 18: tile 16 reading the weight pre-packed tile-major (`_w_tilemajor`): linear 1-KiB operand requests


def _decode_arm(tile: int, split_k: int):
    """autotune arm id -> (C-ABI tile id, split_k).  Ids 16..127 encode split-K (geometry + 16 log2(split)), 128+ / 256+ stream-K and its
    hybrid; ARM_160 (512) is the C-ABI tile 16, the 160 x 320 kernel."""
    if ARM_SMALLM <= tile < ARM_SMALLM + 4:            # 600 .. 603: the small-M tiles of 32 x 64 per wave (C-ABI tiles 19 .. 22)
        return 19 + tile - ARM_SMALLM, 1
    if tile == ARM_256:
        return 17, 1
    if ARM_160B <= tile < ARM_160B + 5:                # 544 + log2(split): the same on the tile-major weight copy
        return 18, 1 << (tile - ARM_160B)
    if ARM_160 <= tile < ARM_160 + 5:                  # 512 + log2(split): the 160 x 320 kernel, split-K 1 / 2 / 4 / 8 / 16
        return 16, 1 << (tile - ARM_160)
    if tile >= 384:
        return tile - 384, -3                           # k-lockstep split on the 8-phase arms (round 4): chunks of every tile's reduction, an XCD's CUs on the same chunk
    if tile >= 256:
        return tile - 256, -2                           # whole rounds on the plain grid, the last partial round stream-K (8-phase arms)
    if tile >= 128:
        return tile - 128, -1                           # stream-K
    if tile >= 16:
        return tile & 15, 1 << (tile >> 4)
    return tile, split_k


_sk_ws = {}


def _streamk_workspace(device):
    """[4 KiB of flags | fp32 partial tiles] for the stream-K arms: zero on first use, handed back zeroed by the kernel.
    One per (device, stream): two streams running stream-K launches at once must not share flags / partial slots."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _sk_ws.get(key)
    if ws is None:
        # (8-phase arms: <= 7 slots per CU.)  Only the flag words must start at zero; the partial slots are written before they are
        # read.  A full torch.zeros here was a 486 MB fill -- and, first reached under graph capture (the capture stream is a new
        # key), one that was replayed with every step: 60 us of the 16x320x512 step.
        ws = torch.empty((4096 + 1856 * 256 * 256 * 4) // 4, dtype=torch.float32, device=device)
        ws[:1024].zero_()
        _sk_ws[key] = ws
    return ws.data_ptr(), ws.numel() * 4


def prepare_streamk_workspace(stream) -> None:
    """Create and zero the stream-K workspace of `stream` EAGERLY.  Call before capturing a HIP graph on that stream: first reached under
    capture, the flag words' zero fill is only recorded into THAT graph, and a second graph captured on the same stream and replayed first
    would run the flag-based stream-K arms on uninitialised flags."""
    with torch.cuda.stream(stream):
        _streamk_workspace(torch.device("cuda", torch.cuda.current_device()))
    stream.synchronize()


def release_streamk_workspace(stream) -> None:
    """Drop the workspace of a stream that is going away (a discarded graph runner): 486 MB per key otherwise stay allocated."""
    dev = getattr(getattr(stream, "device", None), "index", None)          # the stream's own device, not whichever is current when a runner dies
    _sk_ws.pop((torch.cuda.current_device() if dev is None else dev, stream.cuda_stream), None)


def _splitk_workspace(device, split_k: int, M: int, N: int):
    if split_k < 0:                                     # -1 / -2 / -3 / <= -16: the stream-K forms
        return _streamk_workspace(device)
    if split_k <= 1:
        return None, 0
    nbytes = split_k * M * N * 4
    return _workspace(device, nbytes).data_ptr(), nbytes


def split_arms(M: int, N: int, Kd: int):
    """extra autotune arms for problems whose 128x128 output tiles cannot fill the 256 CUs: split-K 2 / 4 (/ 8)"""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles > 320 or Kd < 1024:
        return ()
    arms = [t + 16 * si for si in (1, 2) for t in (1, 2, 9)]
    if tiles <= 128 and Kd >= 4096:
        arms += [1 + 16 * 3, 9 + 16 * 3]
    if N % 320 == 0:                                    # 160 x 320 tiles with split-K: as many workgroups as CUs, whole rounds
        t160 = ((M + 159) // 160) * (N // 320)
        arms += [(ARM_160B if W_TILEMAJOR else ARM_160) + si for si in (1, 2, 3, 4) if 128 <= t160 * (1 << si) <= 512 and Kd // 32 >= 8 * (1 << si)]
    return tuple(arms)


W_TILEMAJOR = os.environ.get("FMC_W_TILEMAJOR", "1") != "0"      # A/B switch: the 160 x 320 kernels read weights pre-packed tile-major (tile 18)


def _owner_cache(t: torch.Tensor, name: str) -> dict:
    """A dict living on the tensor that owns t's storage (dies with it), reset when that tensor's version changes."""
    owner = t._base if t._base is not None else t
    cache = getattr(owner, name, None)
    if cache is None or cache[0] != owner._version:
        cache = (owner._version, {})
        try:
            setattr(owner, name, cache)
        except Exception:
            pass
    return cache[1]


def _w_tilemajor(weight: torch.Tensor) -> torch.Tensor:
    """`[N, K]` projection weight -> `[N / 320][K / 32][320][32]`: what tile 18 reads (a W piece of a sub-tile = one contiguous KiB)."""
    cache = _owner_cache(weight, "_fmc_wtm")
    key = ("lin", weight.storage_offset(), tuple(weight.shape), tuple(weight.stride()), weight._version)
    hit = cache.get(key)
    if hit is None:
        N, Kd = weight.shape
        with torch.no_grad():
            hit = weight.detach().reshape(N // 320, 320, Kd // 32, 32).permute(0, 2, 1, 3).contiguous()
        cache[key] = hit
    return hit


def _w_tilemajor_conv(weight_cl: torch.Tensor) -> torch.Tensor:
    """Channels-last 3x3 filter (physically `[Cout][3][3][Cin]`) -> `[Cout / 320][Cin / 64][9 taps][2 halves][320][32]`: the conv kernel's own
    sub-tile order (64-channel chunk, tap, 32-channel half)."""
    cache = _owner_cache(weight_cl, "_fmc_wtm")
    key = ("conv", weight_cl.storage_offset(), tuple(weight_cl.shape), tuple(weight_cl.stride()), weight_cl._version)
    hit = cache.get(key)
    if hit is None:
        cout, cin = weight_cl.shape[:2]
        with torch.no_grad():
            w = weight_cl.detach().permute(0, 2, 3, 1).reshape(cout // 320, 320, 9, cin // 64, 2, 32)      # [nt][row][tap][chunk][half][32]
            hit = w.permute(0, 3, 2, 4, 1, 5).contiguous()                                                 # [nt][chunk][tap][half][row][32]
        cache[key] = hit
    return hit


def pack_temporal_qkv(w_qkv: torch.Tensor, heads: int = 8) -> torch.Tensor:
    """Fused temporal `[3 C, C]` projection (rows q | k | v, heads contiguous inside each) -> the MFMA-fragment order of `fmc_temporal_block_bf16`
    (C = 320, d = 40): per head [q: blocks q[0:16], q[16:32], tail = (q[32:40] | k[32:40])][k: k[0:16], k[16:32]][v: v[0:16], v[16:32], (v[32:40] | 0)],
    every part as [k-step C / 32][block][lane = 16 * (k chunk of 8) + row][8]: the fragment of one 16-row block and k-step is one contiguous KiB,
    a head's 30 (part, k-step) steps one contiguous 80-KiB stream."""
    C3, C = w_qkv.shape
    assert C3 == 3 * C and C == 320 and heads == 8, "the fused temporal block exists for C = 320, 8 heads"
    d = C // heads
    w = w_qkv.detach()
    q, k, v = w[:C].view(heads, d, C), w[C:2 * C].view(heads, d, C), w[2 * C:].view(heads, d, C)
    z = torch.zeros(heads, 8, C, dtype=w.dtype, device=w.device)
    parts = [torch.stack([q[:, 0:16], q[:, 16:32], torch.cat([q[:, 32:40], k[:, 32:40]], 1)], 1),        # [H, 3, 16, C]
             torch.stack([k[:, 0:16], k[:, 16:32]], 1),
             torch.stack([v[:, 0:16], v[:, 16:32], torch.cat([v[:, 32:40], z], 1)], 1)]
    out = []
    for blk in parts:                                   # [H, nb, 16 rows, C] -> [H, ks, nb, kq, row, 8]
        nb = blk.shape[1]
        out.append(blk.reshape(heads, nb, 16, C // 32, 4, 8).permute(0, 3, 1, 4, 2, 5).reshape(heads, -1))
    return torch.cat(out, 1).contiguous().view(-1)


def pack_w_frag80(w: torch.Tensor) -> torch.Tensor:
    """`[640, 640]` projection weight -> the MFMA-fragment order of the C = 640 fused temporal block (`temporal_block640.hip`): wave w owns output columns
    80 w .. 80 w + 79; per wave [k-step K / 32][block of 16 rows][lane = 16 * (k chunk of 8) + row][8]: the fragment of one block and k-step is one contiguous
    KiB, a wave's 100 (k-step, block) fragments one contiguous 100-KiB stream that never passes through LDS."""
    N, Kd = w.shape
    assert N == 640 and Kd == 640, "the C = 640 fused temporal block"
    return w.detach().reshape(8, 5, 16, Kd // 32, 4, 8).permute(0, 3, 1, 4, 2, 5).contiguous().view(-1)


def pack_temporal_qkv80(w_qkv: torch.Tensor, heads: int = 8) -> torch.Tensor:
    """Fused temporal `[3 C, C]` projection at C = 640 (d = 80 = five whole 16-row blocks per head and part, no padding): per head [q | k | v], every part
    as [k-step][block][lane][8] like `pack_w_frag80`."""
    C3, C = w_qkv.shape
    assert C3 == 3 * C and C == 640 and heads == 8, "the C = 640 fused temporal block: 8 heads x 80"
    w = w_qkv.detach()
    parts = [w[i * C:(i + 1) * C].reshape(heads, 5, 16, C // 32, 4, 8).permute(0, 3, 1, 4, 2, 5).reshape(heads, -1) for i in range(3)]
    return torch.cat(parts, 1).contiguous().view(-1)


def temporal_block_supported(h: torch.Tensor, heads: int) -> bool:
    """`[B, F, hw, C]` bf16 tokens the fused block takes: F = 16, 8 heads, C = 320 with hw % 10 == 0 or C = 640 with hw % 5 == 0, inference."""
    return (h.is_cuda and h.dtype == torch.bfloat16 and h.ndim == 4 and h.is_contiguous() and h.shape[1] == 16 and heads == 8
            and ((h.shape[3] == 320 and h.shape[2] % 10 == 0) or (h.shape[3] == 640 and h.shape[2] % 5 == 0 and TEMPORAL_FUSED_640))
            and h.numel() * 2 < (1 << 31) and not torch.is_grad_enabled())


TEMPORAL_FUSED_640 = os.environ.get("FMC_TEMPORAL_FUSED_640", "1") != "0"      # A/B switch: the 20x32-level blocks on the un-fused chain


def temporal_block(h: torch.Tensor, ln_gamma: torch.Tensor, ln_bpe: torch.Tensor, ln_eps: float, w_qkv_packed: torch.Tensor,
                   w_out_tm: torch.Tensor, b_out: Optional[torch.Tensor], scale: float, w_merge_tm: Optional[torch.Tensor] = None,
                   pose_term: Optional[torch.Tensor] = None, merge_scale: float = 1.0, stats_eps: Optional[float] = None):
    """One attention block of the temporal transformer in one launch (`fmc_temporal_block_bf16`): LayerNorm (+ pe) -> [Camera-Adapter merge] ->
    q | k | v -> attention over the frames -> out-projection + bias + h.  Returns `out` or `(out, ln_stats)` when `stats_eps` is given.
    C = 320: `w_qkv_packed = pack_temporal_qkv(..)`, `w_out_tm` / `w_merge_tm` = `_w_tilemajor(..)`.  C = 640: `pack_temporal_qkv80`, `pack_w_frag80`, no statistics."""
    _dev(h, ln_gamma, ln_bpe, w_qkv_packed, w_out_tm, b_out, w_merge_tm, pose_term)
    B, F, hw, C = h.shape
    assert h.is_contiguous() and ln_gamma.dtype == torch.float32 and ln_bpe.dtype == torch.float32 and tuple(ln_bpe.shape) == (F, C) and ln_bpe.is_contiguous()
    assert (w_merge_tm is None) == (pose_term is None)
    assert pose_term is None or (pose_term.shape == h.shape and pose_term.is_contiguous() and pose_term.dtype == h.dtype)
    out = torch.empty_like(h)
    stats = torch.empty(B * F * hw, 2, dtype=torch.float32, device=h.device) if stats_eps is not None else None
    _log_call("fused_block", ("temporal", B * F * hw, C, w_merge_tm is not None),
              2.0 * B * F * hw * C * C * (5 if w_merge_tm is not None else 4) + 4.0 * B * F * hw * F * C)
    _lib.check(_lib.load().fmc_temporal_block_bf16(h.data_ptr(), out.data_ptr(), ln_gamma.data_ptr(), ln_bpe.data_ptr(), float(ln_eps), _p(w_merge_tm),
                                                   _p(pose_term), float(merge_scale), w_qkv_packed.data_ptr(), w_out_tm.data_ptr(), _p(b_out),
                                                   _p(stats), float(stats_eps or 0.0), B, F, hw, C, 8, float(scale), _stream()),
               "fmc_temporal_block_bf16")
    return out if stats is None else (out, stats)


XATTN_FUSED_640 = os.environ.get("FMC_XATTN_FUSED_640", "1") != "0"      # A/B switch: the 20x32-level text cross-attention as one launch
XATTN_FUSED_320 = os.environ.get("FMC_XATTN_FUSED_320", "1") != "0"      # ... and the 40x64-level one


def pack_xattn_q40(w_q: torch.Tensor, heads: int = 8) -> torch.Tensor:
    """`to_q [320, 320]` of a text cross-attention (8 heads x 40) -> the fragment order phase D of the fused block reads: per head [10 k-steps][3 blocks =
    q[0:16], q[16:32], (q[32:40] | 8 zero rows)][lane = 16 * (k chunk of 8) + row][8] (the q part of `pack_temporal_qkv`, the shared tail block's k half empty)."""
    C, Ci = w_q.shape
    assert C == 320 and Ci == 320 and heads == 8
    q = w_q.detach().view(heads, 40, Ci)
    z = torch.zeros(heads, 8, Ci, dtype=q.dtype, device=q.device)
    blk = torch.stack([q[:, 0:16], q[:, 16:32], torch.cat([q[:, 32:40], z], 1)], 1)            # [H, 3, 16, C]
    return blk.reshape(heads, 3, 16, Ci // 32, 4, 8).permute(0, 3, 1, 4, 2, 5).contiguous().view(-1)


def xattn_block_supported(h: torch.Tensor, kv_tokens: int, heads: int) -> bool:
    """`[images, hw, C]` bf16 tokens the fused text cross-attention block takes: 8 heads, at most 80 text tokens, inference; C = 640 with hw % 80 == 0
    or C = 320 with hw % 160 == 0."""
    if not (h.is_cuda and h.dtype == torch.bfloat16 and h.ndim == 3 and h.is_contiguous() and heads == 8 and 0 < kv_tokens <= 80
            and h.numel() * 2 < (1 << 31) and not torch.is_grad_enabled()):
        return False
    return (XATTN_FUSED_640 and h.shape[2] == 640 and h.shape[1] % 80 == 0) or (XATTN_FUSED_320 and h.shape[2] == 320 and h.shape[1] % 160 == 0)


xattn_block640_supported = xattn_block_supported


def xattn_pack_kv(kv: torch.Tensor) -> torch.Tensor:
    """`kv [B, S, 2 C]` (the text's fused k | v projection, C = 320 | 640, S <= 80) -> the K / V^T MFMA fragments the fused text cross-attention block
    reads (`fmc_xattn_pack_kv40` / `fmc_xattn_pack_kv`).  Constant over the denoising steps of a clip: packed once per clip (`Attention.text_kv`)."""
    _dev(kv)
    B, S, C2 = kv.shape
    C = C2 // 2
    assert C in (320, 640) and kv.stride(2) == 1 and kv.stride(1) == C2 and S <= 80
    L = _lib.load()
    if C == 640:
        frag = torch.empty(B * 8 * 12800, dtype=kv.dtype, device=kv.device)
        _lib.check(L.fmc_xattn_pack_kv(kv.data_ptr(), frag.data_ptr(), B, S, kv.stride(0), _stream()), "fmc_xattn_pack_kv")
    else:
        frag = torch.empty(B * 8 * 7680, dtype=kv.dtype, device=kv.device)
        _lib.check(L.fmc_xattn_pack_kv40(kv.data_ptr(), frag.data_ptr(), B, S, kv.stride(0), _stream()), "fmc_xattn_pack_kv40")
    return frag


def xattn_block(h: torch.Tensor, ln_gamma: torch.Tensor, ln_btab: torch.Tensor, ln_eps: float, w_q_packed: torch.Tensor, kv: torch.Tensor,
                w_out_packed: torch.Tensor, b_out: Optional[torch.Tensor], scale: float, images_per_text: int, stats_eps: Optional[float] = None,
                frag: Optional[torch.Tensor] = None):
    """`to_out(softmax(to_q(LayerNorm(h)) k^T scale) v) + b + h` in one launch; `kv [B, S, 2 C]` = the text's fused k | v projection (packed into MFMA
    fragments by one tiny launch); `ln_btab [16, C]` fp32 rows = the LayerNorm beta.  C = 640: `fmc_xattn_block640_bf16`, weights `pack_w_frag80`;
    C = 320: `fmc_xattn_block320_bf16`, `pack_xattn_q40` / `_w_tilemajor`, and `stats_eps` adds the (mean, rstd) of the output rows: `(out, stats)`."""
    _dev(h, ln_gamma, ln_btab, w_q_packed, kv, w_out_packed, b_out)
    N, hw, C = h.shape
    B, S, C2 = kv.shape
    assert C in (320, 640) and C2 == 2 * C and kv.stride(2) == 1 and kv.stride(1) == C2 and N % images_per_text == 0 and N // images_per_text == B
    out = torch.empty_like(h)
    L = _lib.load()
    _log_call("fused_block", ("xattn", N * hw, C, S), 2.0 * N * hw * C * C * 2 + 4.0 * N * hw * S * C)
    if frag is None:
        frag = xattn_pack_kv(kv)
    assert frag.numel() == B * 8 * (12800 if C == 640 else 7680) and frag.dtype == h.dtype
    if C == 640:
        assert stats_eps is None
        _lib.check(L.fmc_xattn_block640_bf16(h.data_ptr(), out.data_ptr(), ln_gamma.data_ptr(), ln_btab.data_ptr(), float(ln_eps), w_q_packed.data_ptr(),
                                             frag.data_ptr(), w_out_packed.data_ptr(), _p(b_out), N, hw, S, images_per_text, float(scale), _stream()),
                   "fmc_xattn_block640_bf16")
        return out
    stats = torch.empty(N * hw, 2, dtype=torch.float32, device=h.device) if stats_eps is not None else None
    _lib.check(L.fmc_xattn_block320_bf16(h.data_ptr(), out.data_ptr(), ln_gamma.data_ptr(), ln_btab.data_ptr(), float(ln_eps), w_q_packed.data_ptr(),
                                         frag.data_ptr(), w_out_packed.data_ptr(), _p(b_out), _p(stats), float(stats_eps or 0.0), N, hw, S, images_per_text,
                                         float(scale), _stream()), "fmc_xattn_block320_bf16")
    return out if stats is None else (out, stats)


xattn_block640 = xattn_block


GEGLU_DIRECT_640 = os.environ.get("FMC_GEGLU_DIRECT_640", "1") != "0"      # A/B switch: LayerNorm + GEGLU projection of the 20x32 level with the A operand resident
GEGLU_DIRECT_320 = os.environ.get("FMC_GEGLU_DIRECT_320", "1") != "0"      # ... and of the 40x64 level


def pack_geglu_frag80(w: torch.Tensor) -> torch.Tensor:
    """GEGLU projection `[2 Cff, C]` (value rows, then gate rows; C = 640 | 320) -> `fmc_geglu640_ln_bf16` / `fmc_geglu320_ln_bf16`'s fragment order:
    [Cff / (40 NW) chunks][NW = C / 80 waves][C / 32 k-steps][5 blocks][lane][8]; wave w of chunk c owns gated columns 40 NW c + 40 w .. + 39, its 80 weight
    rows ordered [v 0-15 | v 16-31 | v 32-39, g 32-39 | g 0-15 | g 16-31] so that value and gate of a column share a lane in four of the five blocks."""
    two_cff, Kd = w.shape
    cff, nw = two_cff // 2, Kd // 80
    assert Kd in (320, 640) and cff % (40 * nw) == 0
    nc = cff // (40 * nw)
    base = (torch.arange(nc, device=w.device)[:, None] * (40 * nw) + torch.arange(nw, device=w.device)[None, :] * 40)[..., None]      # [nc, nw, 1]
    r = torch.arange(16, device=w.device)
    idx = torch.cat([base + r, base + 16 + r, base + 32 + r[:8], cff + base + 32 + r[:8], cff + base + r, cff + base + 16 + r], -1)    # [nc, nw, 80]
    wp = w.detach()[idx.reshape(-1)].reshape(nc, nw, 5, 16, Kd // 32, 4, 8)
    return wp.permute(0, 1, 4, 2, 5, 3, 6).contiguous().view(-1)


GEGLU_PIPE = os.environ.get("FMC_GEGLU_PIPE", "1") != "0"        # A/B switch: the software-pipelined LayerNorm + GEGLU kernel (csrc/geglu_pipe.hip) where it applies
GEGLU_PIPE_640 = os.environ.get("FMC_GEGLU_PIPE_640", "0") == "1"  # ... also at C = 640 (80-row form only: 130 vs 136 us isolated)


def pack_geglu_frag(w: torch.Tensor, group: int = 32) -> torch.Tensor:
    """GEGLU projection `[2 Cff, C]` (value rows, then gate rows) -> `fmc_geglu_pipe_ln_bf16`'s fragment order: per group of `group` (32 | 16) gated columns
    the 2 x group weight rows [value blocks of 16 | gate blocks of 16] (value and gate of a column share a lane), as
    [Cff / group][C / 32 k-steps][group / 8 blocks][lane = 16 kq + row][8]."""
    two_cff, Kd = w.shape
    cff, nb = two_cff // 2, group // 8
    assert Kd % 32 == 0 and cff % group == 0 and group in (16, 32)
    base = (torch.arange(cff // group, device=w.device) * group)[:, None]                                                            # [ng, 1]
    r = torch.arange(group, device=w.device)
    idx = torch.cat([base + r, cff + base + r], -1)                                                                                  # [ng, 2 group]
    wp = w.detach()[idx.reshape(-1)].reshape(cff // group, nb, 16, Kd // 32, 4, 8)                                                   # [ng, block, row, ks, kq, 8]
    return wp.permute(0, 3, 1, 4, 2, 5).contiguous().view(-1)


def geglu_pipe_variant(M: int, C: int) -> int:
    """1 = the 160-row form (C = 320, M % 160 == 0: each weight fragment serves twice the rows), else 0 = the 80-row form.  FMC_GEGLU_PIPE_VARIANT forces."""
    env = os.environ.get("FMC_GEGLU_PIPE_VARIANT")
    if env is not None:
        return int(env)
    return 1 if (C == 320 and M % 160 == 0) else 0


def geglu_ln_pipe_ok(h: torch.Tensor, weight: torch.Tensor) -> bool:
    C = h.shape[-1]
    M = h.numel() // C
    return bool(GEGLU_PIPE and h.is_cuda and h.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and h.is_contiguous() and weight.shape[1] == C
                and not torch.is_grad_enabled()
                and _lib.load().fmc_geglu_pipe_supported(M, weight.shape[0] // 2, C, geglu_pipe_variant(M, C)))


def geglu_ln_pipe(h: torch.Tensor, ln_gamma: torch.Tensor, ln_beta: torch.Tensor, ln_eps: float, w_packed: torch.Tensor, bias: Optional[torch.Tensor],
                  cff: int, blocked: bool = False, variant: int = 0) -> torch.Tensor:
    """`GEGLU(LayerNorm(h))` on `fmc_geglu_pipe_ln_bf16` (the gate of chunk c - 1 in the shadow of chunk c's MFMAs): `[..., C] -> [..., cff]`, `w_packed` =
    `pack_geglu_frag(weight, 32 if variant == 0 else 16)`.  `blocked`: tile-major `[M / 160][cff / 32][160][32]` for `linear_from_blocked`."""
    _dev(h, ln_gamma, ln_beta, w_packed, bias)
    C = h.shape[-1]
    M = h.numel() // C
    out = torch.empty(*h.shape[:-1], cff, dtype=h.dtype, device=h.device)
    _log_call("geglu_direct", (M, 2 * cff, C), 2.0 * M * 2 * cff * C)
    _lib.check(_lib.load().fmc_geglu_pipe_ln_bf16(h.data_ptr(), out.data_ptr(), ln_gamma.data_ptr(), ln_beta.data_ptr(), float(ln_eps), w_packed.data_ptr(),
                                                  _p(bias), M, cff, C, int(blocked), int(variant), _stream()), "fmc_geglu_pipe_ln_bf16")
    return out


def geglu_ln_direct_ok(h: torch.Tensor, weight: torch.Tensor) -> bool:
    C = h.shape[-1]
    if not (h.is_cuda and h.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and h.is_contiguous() and weight.shape[1] == C
            and (h.numel() // C) % 80 == 0 and h.numel() * 2 < (1 << 31) and not torch.is_grad_enabled()):
        return False
    return (GEGLU_DIRECT_640 and C == 640 and (weight.shape[0] // 2) % 320 == 0) or (GEGLU_DIRECT_320 and C == 320 and (weight.shape[0] // 2) % 160 == 0)


GEGLU_DIRECT_BLOCKED = os.environ.get("FMC_GEGLU_DIRECT_BLOCKED", "1") != "0"   # A/B switch: the level-0 feed-forward's intermediate tile-major (below)


def geglu_direct_blocked_ok(h: torch.Tensor, w2: torch.Tensor, residual) -> bool:
    """Level 0 (C = 320): `geglu_ln_direct(..., blocked=True)` + `linear_from_blocked` -- the second GEMM (K = 1280, N = 320) streams a 210-MB operand in
    128-byte row pieces 2560 bytes apart when it is row-major; tile-major it requests contiguous 10-KiB blocks (113 -> 99 us isolated,
    tools/scratch/r05/bench_ffblk.py).  No gain measured at C = 640 (74.6 / 75.2 us), so only this level takes it."""
    C = h.shape[-1]
    M = h.numel() // C
    N2, Cff = w2.shape
    return (GEGLU_DIRECT_BLOCKED and FF_BLOCKED and C == 320 and M % 160 == 0 and N2 % 320 == 0 and Cff % 32 == 0 and (M // 160) * (N2 // 320) > _cus(h.device)
            and M * Cff * 2 < (1 << 31) and w2.dtype == torch.bfloat16 and w2.is_contiguous()
            and (residual is None or (residual.is_contiguous() and residual.dtype == h.dtype)) and os.environ.get("FMC_G160_PERSIST", "1") != "0")


def geglu_ln_direct(h: torch.Tensor, ln_gamma: torch.Tensor, ln_beta: torch.Tensor, ln_eps: float, w_packed: torch.Tensor, bias: Optional[torch.Tensor],
                    cff: int, blocked: bool = False) -> torch.Tensor:
    """`GEGLU(LayerNorm(h))` in one launch (`fmc_geglu640_ln_bf16` / `fmc_geglu320_ln_bf16`): `[..., C] -> [..., cff]`.  `blocked`: the result is laid
    out tile-major `[M / 160][cff / 32][160][32]` for `linear_from_blocked` (same shape, private to the feed-forward)."""
    _dev(h, ln_gamma, ln_beta, w_packed, bias)
    C = h.shape[-1]
    M = h.numel() // C
    out = torch.empty(*h.shape[:-1], cff, dtype=h.dtype, device=h.device)
    _log_call("geglu_direct", (M, 2 * cff, C), 2.0 * M * 2 * cff * C)
    fn = _lib.load().fmc_geglu640_ln_bf16 if C == 640 else _lib.load().fmc_geglu320_ln_bf16
    _lib.check(fn(h.data_ptr(), out.data_ptr(), ln_gamma.data_ptr(), ln_beta.data_ptr(), float(ln_eps), w_packed.data_ptr(), _p(bias), M, cff, int(blocked),
                  _stream()), "fmc_geglu_ln_bf16")
    return out


def linear_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.stride(-1) == 1
            and weight.shape[1] % 64 == 0 and weight.shape[0] % 8 == 0 and weight.is_contiguous())


def _rows2d(t: torch.Tensor):
    """(M, row stride) of a tensor whose leading dims collapse to uniformly strided rows with a dense last dim."""
    C = t.shape[-1]
    if t.is_contiguous():
        return t.numel() // C, C
    assert t.ndim == 2 and t.stride(1) == 1, "strided operands must be 2-D [M, C] views"
    return t.shape[0], t.stride(0)


VENDOR_DIRECT = os.environ.get("FMC_VENDOR_DIRECT", "1") != "0"      # A/B switch: the vendor arm through fmc_vendor_linear_bf16 (bias + residual in the GEMM)
VENDOR_ALGO = int(os.environ.get("FMC_VENDOR_ALGO", "-1"))           # >= 0: this heuristic candidate for every problem (clamped to the list); default: per problem, from the arm table
_vendor_seen = {}                                                     # problem -> candidate count (a first call plans: never under stream capture)
vendor_direct_calls = {"direct": 0, "with_residual": 0}


_vendor_ws = {}                                                       # (device, stream handle) -> workspace tensor: two streams never share split-K scratch
_vendor_inited = set()


def _vendor_init(device_index: int) -> None:
    """hipBLASLt handle of the device, created once through the ABI's explicit `fmc_vendor_init` and dropped at interpreter exit."""
    if device_index not in _vendor_inited:
        _lib.check(_lib.load().fmc_vendor_init(), "fmc_vendor_init")
        if not _vendor_inited:
            atexit.register(_vendor_shutdown)
        _vendor_inited.add(device_index)


def _vendor_shutdown() -> None:
    _vendor_ws.clear()
    try:
        if _vendor_inited and torch.cuda.is_available():
            cur = torch.cuda.current_device()
            for d in sorted(_vendor_inited):
                torch.cuda.set_device(d)
                _lib.load().fmc_vendor_destroy()
            torch.cuda.set_device(cur)
    except Exception:                                                 # interpreter teardown: the driver frees what is left
        pass
    _vendor_inited.clear()


def _vendor_workspace(device: torch.device):
    """The caller-owned scratch `fmc_vendor_linear_bf16` wants: one torch allocation per (device, stream).  Under stream capture the buffer comes out
    of the graph's private pool and lives as long as the graph's memory does; the entry is keyed by the capture stream, which eager work never uses."""
    key = (device.index, _stream())
    ws = _vendor_ws.get(key)
    if ws is None:
        ws = _vendor_ws[key] = torch.empty(int(_lib.load().fmc_vendor_workspace_bytes()), dtype=torch.uint8, device=device)
    return ws


def vendor_version() -> int:
    """hipBLASLt's version number (the `("valgo", ...)` entries of an arm table index ITS heuristic's candidate list)."""
    _vendor_init(torch.cuda.current_device())
    return int(_lib.load().fmc_vendor_version())


def vendor_linear_ok(x: torch.Tensor, weight: torch.Tensor, bias, residual) -> bool:
    if not (VENDOR_DIRECT and x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and weight.is_contiguous() and x.stride(-1) == 1
            and (x.is_contiguous() or x.ndim == 2) and (bias is None or (bias.dtype == torch.bfloat16 and bias.is_contiguous()))
            and (residual is None or (residual.dtype == torch.bfloat16 and residual.stride(-1) == 1 and (residual.is_contiguous() or residual.ndim == 2)))
            and not torch.is_grad_enabled()):
        return False
    N, Kd = weight.shape
    M, ldx = _rows2d(x)
    ldres = 0
    if residual is not None:
        # the library reads C as a full [M, N] matrix: a broadcastable residual ([1, S, N], [N]) must take the torch path, which broadcasts
        if tuple(residual.shape) != (*x.shape[:-1], N):
            return False
        ldres = _rows2d(residual)[1]
        if ldres % 8 or residual.data_ptr() % 16:
            return False
    if ldx % 8 or x.data_ptr() % 16 or weight.data_ptr() % 16 or (bias is not None and bias.data_ptr() % 2):
        return False                                                  # un-aligned views: F.linear accepts them, the direct call would raise FMC_E_ALIGN
    key = (x.device.index, M, N, Kd, ldx, ldres, bias is not None, residual is not None)
    return key in _vendor_seen or not torch.cuda.is_current_stream_capturing()


def vendor_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`x @ weight^T + bias + residual` as one hipBLASLt launch (`fmc_vendor_linear_bf16`): the library arm of `linear` without the separate torch add."""
    _dev(x, weight, bias, residual)
    N, Kd = weight.shape
    M, ldx = _rows2d(x)
    _vendor_init(x.device.index)
    out = torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
    ws = _vendor_workspace(x.device)
    ldres = 0 if residual is None else _rows2d(residual)[1]
    key = (x.device.index, M, N, Kd, ldx, ldres, bias is not None, residual is not None)
    n = _vendor_seen.get(key)
    if n is None:
        n = _lib.load().fmc_vendor_linear_candidates(M, N, Kd, ldx, ldres, N, int(bias is not None), int(residual is not None))
        if n <= 0:
            _lib.check(n if n < 0 else -1, "fmc_vendor_linear_candidates")
        _vendor_seen[key] = n
    # which of the heuristic's candidates: its first choice is not the fastest on every shape (5120 x 3840 x 1280: 62 us, the second candidate 52 us --
    # tools/scratch/r05/bench_vendor_algos.py: 0.25 ms per step over the ten main shapes).  Chosen once per problem by timing, kept in the arm table
    # (`("valgo", ...)` keys: the tracked default table holds them like the GEMM / conv arms, so a fresh box runs the measured candidates without tuning)
    akey = ("valgo", M, N, Kd, ldx, ldres, bias is not None, residual is not None)
    if not _cache_state["loaded"]:
        load_autotune_table()
    algo = VENDOR_ALGO if VENDOR_ALGO >= 0 else _choice.get(akey)
    if algo is None:
        if AUTOTUNE and n > 1 and not torch.cuda.is_current_stream_capturing():
            times = {}
            for a in range(n):
                call = lambda a=a: _lib.load().fmc_vendor_linear_bf16(x.data_ptr(), weight.data_ptr(), _p(bias), _p(residual), out.data_ptr(), M, N, Kd, ldx,
                                                                      ldres, N, a, ws.data_ptr(), ws.numel(), _stream())
                for _ in range(5):
                    call()
                best = float("inf")
                for _ in range(3):                      # min of 3 event-bracketed bursts
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        call()
                    e1.record()
                    e1.synchronize()
                    best = min(best, e0.elapsed_time(e1) / 10)
                times[a] = round(best, 4)
            algo = min(times, key=times.get)
            _choice[akey] = algo
            _tune_log[akey] = times
            _cache_state["dirty"] = True
        else:
            algo = 0
    _lib.check(_lib.load().fmc_vendor_linear_bf16(x.data_ptr(), weight.data_ptr(), _p(bias), _p(residual), out.data_ptr(), M, N, Kd, ldx, ldres, N,
                                                  min(int(algo), n - 1), ws.data_ptr(), ws.numel(), _stream()), "fmc_vendor_linear_bf16")
    vendor_direct_calls["direct"] += 1
    vendor_direct_calls["with_residual"] += residual is not None
    _log_call("vendor", (M, N, Kd, bias is not None, residual is not None), 2.0 * M * N * Kd)
    return out


def linear4_bf16(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                 alpha: float = 1.0, residual2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`alpha * (x @ weight^T + bias) + residual [+ residual2]` on `fmc_linear4_bf16` (160 x 160 tiles, 4 waves, software-pipelined: the small-M
    projections of the 10x16 / 5x8 levels).  x `[..., K]` (dense last dim, uniformly strided rows), weight `[N, K]` contiguous."""
    _dev(x, weight, bias, residual, residual2)
    N, Kd = weight.shape
    assert weight.is_contiguous() and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
    M, ldx = _rows2d(x)
    out = torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
    ldres = 0
    if residual is not None:
        assert residual.shape == out.shape
        _, ldres = _rows2d(residual)
    if residual2 is not None:
        assert residual is not None and residual2.shape == out.shape and _rows2d(residual2)[1] == ldres
    _log_call("own_linear", (M, N, Kd, "g4", (residual is not None) + (residual2 is not None)), 2.0 * M * N * Kd)
    _lib.check(_lib.load().fmc_linear4_bf16(x.data_ptr(), weight.data_ptr(), _p(bias), _p(residual), _p(residual2), out.data_ptr(), M, N, Kd, ldx, ldres,
                                            N, float(alpha), _stream()), "fmc_linear4_bf16")
    return out


def linear_bf16(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                residual: Optional[torch.Tensor] = None, alpha: float = 1.0, geglu: bool = False,
                tile: int = 0, split_k: int = 1, x2: Optional[torch.Tensor] = None,
                residual2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`alpha * (x @ weight^T + bias) + residual [+ residual2]` (or the GEGLU gate, see fmc_linear_bf16) on the bf16 MFMA kernel.
    x `[..., K]`, weight `[N, K]`; residual has the output's shape.  `tile` may also be an autotune arm id
    (`tile + 16 * log2(split_k)`).  `x2 [..., K2]`: the A operand is the concat `[x, x2]` (weight `[N, K + K2]`),
    never materialised."""
    tile, split_k = _decode_arm(tile, split_k)
    _dev(x, weight, bias, residual, x2)
    N, Kd = weight.shape
    if tile == 18:
        M0 = x.numel() // x.shape[-1]
        if x2 is not None or N % 320 or split_k < 1 or M0 * max(Kd, N) * 2 >= (1 << 31) or N * Kd * 2 >= (1 << 31):
            tile = 16                                   # (not a case tile 16 takes on a packed weight: row-major, with tile 16's own fall-backs)
        else:
            weight = _w_tilemajor(weight)
    M, ldx = _rows2d(x)
    ldx2, k_split = 0, 0
    if x2 is not None:
        M2, ldx2 = _rows2d(x2)
        k_split = x.shape[-1]
        assert M2 == M and k_split + x2.shape[-1] == Kd
    n_out = N // 2 if geglu else N
    out = torch.empty(*x.shape[:-1], n_out, dtype=x.dtype, device=x.device)
    ldres = 0
    if residual is not None:
        assert residual.shape == out.shape
        _, ldres = _rows2d(residual)
    if residual2 is not None:
        assert residual is not None and residual2.shape == out.shape and _rows2d(residual2)[1] == ldres
    ws, ws_bytes = _splitk_workspace(x.device, split_k, M, N)
    _log_call("own_linear", (M, N, Kd, "geglu" if geglu else f"t{tile}", (residual is not None) + (residual2 is not None)), 2.0 * M * N * Kd)
    _lib.check(_lib.load().fmc_linear_bf16(x.data_ptr(), weight.data_ptr(), _p(bias), _p(residual), out.data_ptr(), M, N,
                                           Kd, ldx, ldres, n_out, float(alpha), int(geglu), int(tile), int(split_k),
                                           ws, ws_bytes, _p(x2), ldx2, k_split, _p(residual2), _stream()),
               "fmc_linear_bf16")
    return out


def conv3x3_supported(x: torch.Tensor, weight: torch.Tensor, stride, padding) -> bool:
    s2 = tuple(stride) == (2, 2) and x.shape[-1] % 2 == 0 and x.shape[-2] % 2 == 0
    return (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
            and tuple(weight.shape[2:]) == (3, 3) and (tuple(stride) == (1, 1) or s2) and tuple(padding) == (1, 1)
            and weight.shape[1] % 64 == 0 and weight.shape[0] % 8 == 0)


def conv3x3_bf16(x_nhwc: torch.Tensor, weight_cl: torch.Tensor, bias: Optional[torch.Tensor] = None,
                 temb: Optional[torch.Tensor] = None, residual_nhwc: Optional[torch.Tensor] = None,
                 tile: int = 0, split_k: int = 1, temb_div: int = 1, upsample: bool = False,
                 stride2: bool = False) -> torch.Tensor:
    """x `[N, H, W, Cin]` contiguous, weight `[Cout, Cin, 3, 3]` in channels_last memory format (physically
    `[Cout, 3, 3, Cin]`), temb `[N // temb_div, Cout]` (rows may be strided: a column slice of a wider matrix),
    residual `[N, H, W, Cout]` -> `[N, H, W, Cout]`."""
    _dev(x_nhwc, weight_cl, bias, temb, residual_nhwc)
    n, h, w, cin = x_nhwc.shape
    if upsample:                                        # x is the half-resolution source of a nearest 2x upsample
        h, w = 2 * h, 2 * w
    if stride2:                                         # 3x3 / stride 2 / pad 1 (even input size): output is half-size
        assert not upsample and h % 2 == 0 and w % 2 == 0
        h, w = h // 2, w // 2
    cout = weight_cl.shape[0]
    assert x_nhwc.is_contiguous() and weight_cl.is_contiguous(memory_format=torch.channels_last)
    assert temb is None or (temb.stride(1) == 1 and temb.shape == (n // temb_div, cout) and n % temb_div == 0)
    assert residual_nhwc is None or (residual_nhwc.is_contiguous() and residual_nhwc.shape == (n, h, w, cout))
    out = torch.empty(n, h, w, cout, dtype=x_nhwc.dtype, device=x_nhwc.device)
    tile, split_k = _decode_arm(tile, split_k)
    if tile == 18:
        if cout % 320 or cin % 64 or split_k < 1 or x_nhwc.numel() * 2 >= (1 << 31) or weight_cl.numel() * 2 >= (1 << 31):
            tile = 16
        else:
            weight_cl = _w_tilemajor_conv(weight_cl)
    ws, ws_bytes = _splitk_workspace(x_nhwc.device, split_k, n * h * w, cout)
    _lib.check(_lib.load().fmc_conv3x3_bf16(x_nhwc.data_ptr(), weight_cl.data_ptr(), _p(bias), _p(temb),
                                            _p(residual_nhwc), out.data_ptr(), n, h, w, cin, cout,
                                            0 if temb is None else temb.stride(0), int(temb_div), 2 if stride2 else int(upsample),
                                            int(tile),
                                            int(split_k), ws, ws_bytes, _stream()),
               "fmc_conv3x3_bf16")
    return out


LN_EPILOGUE = os.environ.get("FMC_LN_EPILOGUE", "1") != "0"      # A/B switch: the consumer's LayerNorm out of the producing GEMM's epilogue
ln_epilogue_calls = {"emitted": 0, "consumed": 0}


class LnSpec:
    """The LayerNorm a GEMM output feeds (`fmc_linear_bf16_ln`): fp32 gamma / beta, eps, optional positional-encoding table `[>= frames, C]`
    with `pe[(row // pe_inner) % pe_frames]` added after normalising; `key` identifies the consumer (module, pe arguments)."""
    __slots__ = ("gamma", "beta", "eps", "pe", "pe_inner", "pe_frames", "key", "stats_only")

    def __init__(self, gamma, beta, eps, pe, pe_inner, pe_frames, key, stats_only: bool = False):
        self.gamma, self.beta, self.eps, self.pe, self.pe_inner, self.pe_frames, self.key = gamma, beta, eps, pe, pe_inner, pe_frames, key
        self.stats_only = stats_only                    # the consumer is a GEMM that applies the norm itself (`linear_lnc`): only (mean, rstd) per row


_cu_count = {}


def _cus(device) -> int:
    n = _cu_count.get(device.index)
    if n is None:
        n = _cu_count[device.index] = torch.cuda.get_device_properties(device).multi_processor_count & ~7
    return n


def ln_emit_ok(x: torch.Tensor, weight: torch.Tensor, residual, residual2, ln: LnSpec) -> bool:
    N, Kd = weight.shape
    M = x.numel() // x.shape[-1]
    return (LN_EPILOGUE and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
            and x.is_contiguous() and weight.is_contiguous() and N == 320 and Kd % 64 == 0 and M % 160 == 0 and M // 160 > _cus(x.device)
            and ln.gamma.numel() == 320 and ln.gamma.dtype == torch.float32 and ln.beta.dtype == torch.float32
            and (ln.pe is None or (ln.pe.dtype == torch.float32 and ln.pe.is_contiguous() and ln.pe.shape[-1] == 320 and ln.pe_inner % 160 == 0
                                   and ln.pe.shape[0] >= ln.pe_frames))
            and (residual is None or (residual.is_contiguous() and residual.dtype == x.dtype))
            and (residual2 is None or residual2.is_contiguous()) and M * max(Kd, 320) * 2 < (1 << 31) and os.environ.get("FMC_G160_PERSIST", "1") != "0")


def carry_ln(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    tag = getattr(src, "_fmc_ln", None)
    if tag is not None:
        dst._fmc_ln = tag
    return dst


def take_ln(x: torch.Tensor, key) -> Optional[torch.Tensor]:
    """LayerNorm(x) if x's producer already wrote it for exactly this consumer, else None."""
    tag = getattr(x, "_fmc_ln", None)
    if tag is None or tag[1] != key or torch.is_grad_enabled() or tag[2]:
        return None
    ln_epilogue_calls["consumed"] += 1
    return tag[0].view(x.shape)


def take_ln_stats(x: torch.Tensor, key) -> Optional[torch.Tensor]:
    """(mean, rstd) `[M, 2]` fp32 of x's rows if x's producer wrote them for exactly this consumer, else None."""
    tag = getattr(x, "_fmc_ln", None)
    if tag is None or tag[1] != key or torch.is_grad_enabled() or not tag[2]:
        return None
    return tag[0]


def pending_ln(x: torch.Tensor, stats: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> torch.Tensor:
    """x itself (a fresh view, so the tag stays off the residual stream's tensor) marked "LayerNorm still to be applied": the next
    `linear` / `geglu_linear` applies it in its epilogue (`linear_lnc`) or, where it cannot, materialises it first (`resolve_pending_ln`).
    Only call sites whose consumer is one of those two may ask for this (LayerNorm.forward(defer=True))."""
    v = x.view(x.shape)
    v._fmc_pending_ln = (stats, gamma, beta, eps)
    return v


def resolve_pending_ln(x: torch.Tensor) -> torch.Tensor:
    pend = getattr(x, "_fmc_pending_ln", None)
    if pend is None:
        return x
    ln_epilogue_calls["materialised"] = ln_epilogue_calls.get("materialised", 0) + 1
    return _layernorm_raw(x.contiguous(), pend[1], pend[2], pend[3], None, 1, 1)


def _ln_folded_weight(weight: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
    """(W diag(gamma) in bf16, c[n] = sum_k of it, W beta + bias) for `fmc_linear_bf16_lnc`; cached on the tensor that owns W's storage."""
    owner = weight._base if weight._base is not None else weight
    key = (weight.storage_offset(), tuple(weight.shape), weight._version, gamma.data_ptr(), gamma._version, beta.data_ptr(), beta._version,
           None if bias is None else (bias.data_ptr(), bias._version))
    cache = getattr(owner, "_fmc_lnw", None)
    if cache is None or cache[0] != owner._version:
        cache = (owner._version, {})
        try:
            owner._fmc_lnw = cache
        except Exception:
            pass
    hit = cache[1].get(key)
    if hit is None:
        with torch.no_grad():
            w32 = weight.detach().float()
            wg = (w32 * gamma.float()[None, :]).to(torch.bfloat16).contiguous()
            b = w32 @ beta.float()
            if bias is not None:
                b = b + bias.detach().float()
            hit = (wg, wg.float().sum(dim=1).contiguous(), b.contiguous())
        cache[1][key] = hit
    return hit


def lnc_ok(x: torch.Tensor, weight: torch.Tensor) -> bool:
    N, Kd = weight.shape
    M = x.numel() // x.shape[-1]
    return (LN_EPILOGUE and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
            and x.is_contiguous() and N % 320 == 0 and Kd % 64 == 0 and M % 160 == 0 and (M // 160) * (N // 320) > _cus(x.device)
            and M * max(Kd, 320) * 2 < (1 << 31) and N * Kd * 2 < (1 << 31) and os.environ.get("FMC_G160_PERSIST", "1") != "0")


def linear_lnc(x: torch.Tensor, weight: torch.Tensor, bias, pend, geglu: bool = False) -> torch.Tensor:
    """`LayerNorm(x) @ weight^T + bias` (or its GEGLU) with the norm applied in the GEMM's epilogue from the producer's (mean, rstd)."""
    stats, gamma, beta, _ = pend
    wg, c, b = _ln_folded_weight(weight, bias, gamma, beta)
    _dev(x, wg, c, b, stats)
    N, Kd = weight.shape
    M, ldx = _rows2d(x)
    n_out = N // 2 if geglu else N
    out = torch.empty(*x.shape[:-1], n_out, dtype=x.dtype, device=x.device)
    ln_epilogue_calls["consumed"] += 1
    wt = _w_tilemajor(wg) if W_TILEMAJOR else wg
    _log_call("own_linear", (M, N, Kd, "lnc-geglu" if geglu else "lnc", 0), 2.0 * M * N * Kd)
    _lib.check(_lib.load().fmc_linear_bf16_lnc(x.data_ptr(), wt.data_ptr(), out.data_ptr(), M, N, Kd, ldx, n_out, int(geglu), stats.data_ptr(),
                                               c.data_ptr(), b.data_ptr(), int(W_TILEMAJOR), _stream()), "fmc_linear_bf16_lnc")
    return out


def linear_ln(x: torch.Tensor, weight: torch.Tensor, bias, residual, alpha: float, residual2, ln: LnSpec) -> torch.Tensor:
    """`linear_bf16` on the persistent 160 x 320 kernel that also writes the consumer's LayerNorm: out carries `_fmc_ln = (ln_out, key)`."""
    _dev(x, weight, bias, residual, ln.gamma, ln.beta, ln.pe)
    N, Kd = weight.shape
    M, ldx = _rows2d(x)
    out = torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
    ldres = 0 if residual is None else _rows2d(residual)[1]
    ln_epilogue_calls["emitted"] += 1
    if ln.stats_only:
        stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
        ln_out = None
    else:
        stats, ln_out = None, torch.empty_like(out)
    wt = _w_tilemajor(weight) if W_TILEMAJOR else weight
    _log_call("own_linear", (M, N, Kd, "ln", (residual is not None) + (residual2 is not None)), 2.0 * M * N * Kd)
    _lib.check(_lib.load().fmc_linear_bf16_ln(x.data_ptr(), wt.data_ptr(), _p(bias), _p(residual), out.data_ptr(), M, N, Kd, ldx, ldres, N,
                                              float(alpha), _p(residual2), _p(ln_out), ln.gamma.data_ptr(), ln.beta.data_ptr(), float(ln.eps),
                                              _p(ln.pe), int(ln.pe_inner), int(ln.pe_frames), _p(stats), int(W_TILEMAJOR), _stream()),
               "fmc_linear_bf16_ln")
    out._fmc_ln = (stats, ln.key, True) if ln.stats_only else (ln_out, ln.key, False)
    return out


GN_FOLD = os.environ.get("FMC_GN_FOLD", "1") != "0"             # A/B switch: the GroupNorm in front of a transformer's proj_in folded into per-image weights (below)
GN_FOLD_MAX_BYTES = 8 << 20                                       # ... while the per-image weights stay small next to the tensor (level 0: 32 x 200 KB)


def gn_fold_ok(x: torch.Tensor, gn_tag, groups: int, weight: torch.Tensor, ln: Optional[LnSpec]) -> bool:
    """`proj(GroupNorm(x))` without the normalised tensor (`linear_gnfold`): x `[n_img, hw, C]` whose producer left the GroupNorm partial sums (`gn_tag`)."""
    if not GN_FOLD or gn_tag is None or torch.is_grad_enabled() or x.ndim != 3 or weight.ndim != 2:
        return False
    n_img, hw, C = x.shape
    N = weight.shape[0]
    M = n_img * hw
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and weight.dtype == torch.bfloat16 and weight.is_contiguous() and weight.shape[1] == C
            and groups == 32 and gn_tag[1] == C and gn_tag[0].shape[0] == n_img and gn_tag[0].shape[1] <= 64 and gn_tag[0].shape[2] == groups
            and hw % 160 == 0 and C % 64 == 0 and N % 320 == 0 and (M // 160) * (N // 320) >= _cus(x.device) and n_img * N * C * 2 <= GN_FOLD_MAX_BYTES
            and M * max(C, N) * 2 < (1 << 31) and (ln is None or (LN_EPILOGUE and ln.stats_only and N == 320))
            and os.environ.get("FMC_G160_PERSIST", "1") != "0")


def linear_gnfold(x: torch.Tensor, gn_tag, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, weight: torch.Tensor, bias,
                  ln: Optional[LnSpec] = None) -> torch.Tensor:
    """`GroupNorm(x) @ weight^T + bias` with the norm folded into per-image weights: `fmc_groupnorm_fold_linear` turns the producer's partial sums into
    `W'_img = W diag(rstd_img gamma)` (bf16) and an fp32 bias row per image (built from the ROUNDED W': the mean cancels exactly), `fmc_linear_bf16_imgw`
    runs the projection on the raw tensor -- the apply pass (read + write of x) disappears.  `ln` (statistics only): as `linear_ln`."""
    part = gn_tag[0]
    _dev(x, part, gamma, beta, weight, bias)
    n_img, hw, C = x.shape
    N = weight.shape[0]
    M = n_img * hw
    lib = _lib.load()
    w_img = torch.empty(n_img, N, C, dtype=torch.bfloat16, device=x.device)
    b_img = torch.empty(n_img, N, dtype=torch.float32, device=x.device)
    gn_epilogue_calls["consumed"] += 1
    _lib.check(lib.fmc_groupnorm_fold_linear(part.data_ptr(), int(part.shape[1]), gamma.data_ptr(), beta.data_ptr(), weight.data_ptr(), _p(bias),
                                             w_img.data_ptr(), b_img.data_ptr(), n_img, hw, C, int(groups), N, float(eps), int(W_TILEMAJOR), _stream()),
               "fmc_groupnorm_fold_linear")
    out = torch.empty(n_img, hw, N, dtype=x.dtype, device=x.device)
    stats = None
    if ln is not None:
        stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
        ln_epilogue_calls["emitted"] += 1
    _log_call("own_linear", (M, N, C, "gn-fold", 0), 2.0 * M * N * C)
    _lib.check(lib.fmc_linear_bf16_imgw(x.data_ptr(), w_img.data_ptr(), b_img.data_ptr(), out.data_ptr(), M, N, C, C, N, hw, _p(stats),
                                        float(ln.eps) if ln is not None else 0.0, int(W_TILEMAJOR), _stream()), "fmc_linear_bf16_imgw")
    if ln is not None:
        out._fmc_ln = (stats, ln.key, True)
    return out


FF_BLOCKED = os.environ.get("FMC_FF_BLOCKED", "1") != "0"        # A/B switch: tile-major intermediate between the two GEMMs of a feed-forward


def ff_blocked_ok(x: torch.Tensor, w1_il160: Optional[torch.Tensor], w2: torch.Tensor, residual) -> bool:
    """Both GEMMs of the feed-forward on tile 16 with the `[M / 160][Cff / 32][160][32]` intermediate (`fmc_linear_bf16_ffblk`)?"""
    if not FF_BLOCKED or w1_il160 is None or torch.is_grad_enabled():
        return False
    N1, Kd = w1_il160.shape
    N2, Cff = w2.shape
    M = x.numel() // x.shape[-1]
    return (x.is_cuda and x.dtype == torch.bfloat16 and w1_il160.dtype == torch.bfloat16 and w2.dtype == torch.bfloat16 and x.is_contiguous()
            and w1_il160.is_contiguous() and w2.is_contiguous() and N1 == 2 * Cff and Cff % 160 == 0 and N2 % 320 == 0 and Kd % 64 == 0
            and M % 160 == 0 and M >= 16384 and (M // 160) * (N1 // 320) > _cus(x.device) and M * Cff * 2 < (1 << 31)
            and (residual is None or (residual.is_contiguous() and residual.dtype == x.dtype))
            and os.environ.get("FMC_G160_PERSIST", "1") != "0")


def geglu_linear_blocked(x: torch.Tensor, weight_il160: torch.Tensor, bias_il160) -> torch.Tensor:
    """GEGLU projection whose gated output is written tile-major (private to the feed-forward's second GEMM, `linear_from_blocked`).
    A pending LayerNorm on x (LayerNorm.forward(defer=True)) is applied in the epilogue as in `linear_lnc`."""
    pend = getattr(x, "_fmc_pending_ln", None)
    N, Kd = weight_il160.shape
    M = x.numel() // Kd
    out = torch.empty(*x.shape[:-1], N // 2, dtype=x.dtype, device=x.device)
    if pend is not None:
        stats, gamma, beta, _ = pend
        wg, c, b = _ln_folded_weight(weight_il160, bias_il160, gamma, beta)
        _dev(x, wg, c, b, stats)
        ln_epilogue_calls["consumed"] += 1
        wt = _w_tilemajor(wg) if W_TILEMAJOR else wg
        args = (wt.data_ptr(), None, None, out.data_ptr(), M, N, Kd, 0, 1.0, 1, 0, 1, stats.data_ptr(), c.data_ptr(), b.data_ptr(), int(W_TILEMAJOR))
    else:
        _dev(x, weight_il160, bias_il160)
        wt = _w_tilemajor(weight_il160) if W_TILEMAJOR else weight_il160
        args = (wt.data_ptr(), _p(bias_il160), None, out.data_ptr(), M, N, Kd, 0, 1.0, 1, 0, 1, None, None, None, int(W_TILEMAJOR))
    _log_call("own_linear", (M, N, Kd, "geglu-blocked", 0), 2.0 * M * N * Kd)
    _lib.check(_lib.load().fmc_linear_bf16_ffblk(x.data_ptr(), *args, _stream()), "fmc_linear_bf16_ffblk")
    return out


def linear_from_blocked(xb: torch.Tensor, weight: torch.Tensor, bias, residual, alpha: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`alpha * (x @ weight^T + bias) + residual` for an x in the tile-major layout of `geglu_linear_blocked`; `out`: a contiguous `[M, N]` destination."""
    _dev(xb, weight, bias, residual, out)
    N, Kd = weight.shape
    M = xb.numel() // Kd
    if out is None:
        out = torch.empty(*xb.shape[:-1], N, dtype=xb.dtype, device=xb.device)
    assert out.is_contiguous() and out.numel() == M * N and out.dtype == xb.dtype
    wt = _w_tilemajor(weight) if W_TILEMAJOR else weight
    _log_call("own_linear", (M, N, Kd, "from-blocked", int(residual is not None)), 2.0 * M * N * Kd)
    _lib.check(_lib.load().fmc_linear_bf16_ffblk(xb.data_ptr(), wt.data_ptr(), _p(bias), _p(residual), out.data_ptr(), M, N, Kd,
                                                 0 if residual is None else N, float(alpha), 0, 1, 0, None, None, None, int(W_TILEMAJOR), _stream()),
               "fmc_linear_bf16_ffblk")
    return out


FF_TAIL = os.environ.get("FMC_FF_TAIL", "1") != "0"             # A/B switch: feed-forward output projection + proj_out as one product (below)


def ff_tail_ok(h: torch.Tensor, w2: torch.Tensor, wp: torch.Tensor, tail_residual: Optional[torch.Tensor]) -> bool:
    """`proj_out(ff2(g) + b2 + h) + bp + x` as ONE launch on the tile-major intermediate (`ff_tail`): the level-0 transformers, where both GEMMs are
    HBM passes (K = 1280 and K = 320 on 81 920 rows) -- the folded product reads g, h and x once and writes once, the block's output is never stored."""
    C = h.shape[-1]
    M = h.numel() // C
    return (FF_TAIL and not torch.is_grad_enabled() and h.dtype == torch.bfloat16 and h.is_contiguous() and wp.dtype == torch.bfloat16 and w2.dtype == torch.bfloat16
            and wp.ndim == 2 and wp.shape[1] == w2.shape[0] == C and wp.shape[0] % 320 == 0 and C % 64 == 0 and w2.shape[1] % 64 == 0
            and M % 160 == 0 and (M // 160) * (wp.shape[0] // 320) >= _cus(h.device) and M * max(w2.shape[1], C) * 2 < (1 << 31)
            and tail_residual is not None and tail_residual.dtype == h.dtype and tail_residual.is_contiguous()
            and tail_residual.numel() == M * wp.shape[0] and os.environ.get("FMC_G160_PERSIST", "1") != "0")


def fold_ff_tail(w2: torch.Tensor, b2: Optional[torch.Tensor], wp: torch.Tensor, bp: Optional[torch.Tensor]):
    """([Wp W2 | Wp] as bf16 [N, Cff + C], Wp b2 + bp as bf16 [N] or None): folded in fp32, rounded once."""
    with torch.no_grad():
        wpf = wp.detach().float()
        wc = torch.cat([wpf @ w2.detach().float(), wpf], dim=1).to(torch.bfloat16).contiguous()
        bc = None
        if b2 is not None or bp is not None:
            bc = (wpf @ b2.detach().float()) if b2 is not None else torch.zeros(wp.shape[0], dtype=torch.float32, device=wp.device)
            if bp is not None:
                bc = bc + bp.detach().float()
            bc = bc.to(torch.bfloat16).contiguous()
    return wc, bc


def ff_tail(mid_blocked: torch.Tensor, h: torch.Tensor, w_cat: torch.Tensor, b_cat: Optional[torch.Tensor], tail_residual: torch.Tensor,
            gn_hw: int = 0) -> torch.Tensor:
    """`[g | h] @ w_cat^T + b_cat + tail_residual` (`fmc_linear_bf16_fftail`): g = `mid_blocked`, the tile-major gated intermediate; `h` row-major `[M, C]`.
    With `gn_hw` the 160-row tiles also leave the consumer GroupNorm's partial sums (as `linear_gn`).  Returns `[..., N]` in `tail_residual`'s shape."""
    _dev(mid_blocked, h, w_cat, b_cat, tail_residual)
    N, Kd = w_cat.shape
    C = h.shape[-1]
    M = h.numel() // C
    assert mid_blocked.numel() == M * (Kd - C) and tail_residual.numel() == M * N
    out = torch.empty_like(tail_residual)
    part = None
    if gn_hw and gn_emit_ok(M, N, Kd, gn_hw, h.dtype):
        part = torch.empty(M // gn_hw, gn_hw // 160, 32, 2, dtype=torch.float32, device=h.device)
        gn_epilogue_calls["emitted"] += 1
    wt = _w_tilemajor(w_cat) if W_TILEMAJOR else w_cat
    _log_call("own_linear", (M, N, Kd, "ff-tail", 1), 2.0 * M * N * Kd)
    _lib.check(_lib.load().fmc_linear_bf16_fftail(mid_blocked.data_ptr(), h.data_ptr(), wt.data_ptr(), _p(b_cat), tail_residual.data_ptr(), out.data_ptr(),
                                                  M, N, Kd, Kd - C, C, N, _p(part), int(gn_hw if part is not None else 0), int(W_TILEMAJOR), _stream()),
               "fmc_linear_bf16_fftail")
    if part is not None:
        out._fmc_gn = (part, N)
    return out


def linear_gn(x: torch.Tensor, weight: torch.Tensor, bias, residual, alpha: float, residual2, hw: int):
    """`linear_bf16` on the 160 x 320 kernel + the GroupNorm partial sums of the output: (out, partials [M / hw, hw / 160, 32, 2])."""
    _dev(x, weight, bias, residual)
    N, Kd = weight.shape
    M, ldx = _rows2d(x)
    out = torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
    part = torch.empty(M // hw, hw // 160, 32, 2, dtype=torch.float32, device=x.device)
    ldres = 0 if residual is None else _rows2d(residual)[1]
    gn_epilogue_calls["emitted"] += 1
    wt = _w_tilemajor(weight) if W_TILEMAJOR else weight
    _log_call("own_linear", (M, N, Kd, "gn", (residual is not None) + (residual2 is not None)), 2.0 * M * N * Kd)
    _lib.check(_lib.load().fmc_linear_bf16_gn(x.data_ptr(), wt.data_ptr(), _p(bias), _p(residual), out.data_ptr(), M, N, Kd, ldx, ldres, N,
                                              float(alpha), _p(residual2), part.data_ptr(), int(hw), int(W_TILEMAJOR), _stream()), "fmc_linear_bf16_gn")
    out._fmc_gn = (part, N)
    return out


def conv3x3_gn(x_nhwc: torch.Tensor, weight_cl: torch.Tensor, bias, temb, residual_nhwc, temb_div: int, upsample: bool, stride2: bool):
    n, h, w, cin = x_nhwc.shape
    if upsample:
        h, w = 2 * h, 2 * w
    if stride2:
        h, w = h // 2, w // 2
    cout = weight_cl.shape[0]
    out = torch.empty(n, h, w, cout, dtype=x_nhwc.dtype, device=x_nhwc.device)
    part = torch.empty(n, (h * w) // 160, 32, 2, dtype=torch.float32, device=x_nhwc.device)
    gn_epilogue_calls["emitted"] += 1
    wt = _w_tilemajor_conv(weight_cl) if W_TILEMAJOR else weight_cl
    _lib.check(_lib.load().fmc_conv3x3_bf16_gn(x_nhwc.data_ptr(), wt.data_ptr(), _p(bias), _p(temb), _p(residual_nhwc), out.data_ptr(),
                                               n, h, w, cin, cout, 0 if temb is None else temb.stride(0), int(temb_div),
                                               2 if stride2 else int(upsample), part.data_ptr(), int(W_TILEMAJOR), _stream()), "fmc_conv3x3_bf16_gn")
    return out, (part, cout)


# --------------------------------------------------------------------------------------------
# 3x3 convolution with the input halo resident in LDS and GroupNorm + SiLU applied while it is staged (csrc/conv_halo.hip; SURVEY.md
# section 8 f1).  `conv(silu(norm(x)))` of diffusers' ResnetBlock2D as: statistics (out of the producer's epilogue, or one read of x) ->
# `fmc_groupnorm_coef` (per-(image, channel) scale / shift) -> `fmc_conv3x3_halo_bf16` reading the RAW x.  FMC_CONV_HALO=0: A/B switch.
# --------------------------------------------------------------------------------------------
CONV_HALO = os.environ.get("FMC_CONV_HALO", "1") != "0"
CONV_HALO_MIN_TILES = int(os.environ.get("FMC_CONV_HALO_MIN_TILES", "200"))     # workgroups below which the ring / stream-K arms keep the shape
LINEAR4 = os.environ.get("FMC_LINEAR4", "1") != "0"            # A/B switch: `fmc_linear4_bf16` is a candidate arm of the projections
CONV_HALO4 = os.environ.get("FMC_CONV_HALO4", "1") != "0"       # A/B switch: the 4-wave form on the 10x16 / 5x8 levels
CONV_GN_FUSED = os.environ.get("FMC_CONV_GN_FUSED", "0") == "1"   # GroupNorm + SiLU in the conv's operand path instead of a separate apply pass (measured slower)
conv_halo_calls = {"conv": 0, "gn_fused": 0, "stats_pass": 0, "stats_from_producer": 0}


def conv3x3_halo_supported(n: int, h: int, w: int, cin: int, cin1: int, cout: int, upsample: bool) -> bool:
    return bool(_lib.load().fmc_conv3x3_halo_supported(n, h, w, cin, cin1, cout, int(upsample)))


def _w_halo_packed(weight_cl: torch.Tensor) -> torch.Tensor:
    """Channels-last 3x3 filter (physically `[Cout][3][3][Cin]`) -> the halo kernel's sub-tile order (`fmc_conv3x3_halo_pack_weight`), cached on the weight."""
    cache = _owner_cache(weight_cl, "_fmc_wtm")
    key = ("halo", weight_cl.storage_offset(), tuple(weight_cl.shape), tuple(weight_cl.stride()), weight_cl._version)
    hit = cache.get(key)
    if hit is None:
        cout, cin = weight_cl.shape[:2]
        assert weight_cl.is_contiguous(memory_format=torch.channels_last)
        hit = torch.empty(cout * 9 * cin, dtype=weight_cl.dtype, device=weight_cl.device)
        _lib.check(_lib.load().fmc_conv3x3_halo_pack_weight(weight_cl.data_ptr(), hit.data_ptr(), cin, cout, _stream()), "fmc_conv3x3_halo_pack_weight")
        cache[key] = hit
    return hit


def groupnorm_coef(partials: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, hw: int, C: int, groups: int, eps: float,
                   want_stats: bool = False):
    """Partial (sum, sum of squares) `[N, splits, G, 2]` -> per-(image, channel) `(scale, shift)` `[N, C, 2]` of the GroupNorm (+ `[N, G, 2]` mean / rstd)."""
    _dev(partials, gamma, beta)
    N, splits, G, _ = partials.shape
    assert G == groups and partials.dtype == torch.float32 and partials.is_contiguous() and gamma.dtype == torch.float32
    coef = torch.empty(N, C, 2, dtype=torch.float32, device=partials.device)
    stats = torch.empty(N, G, 2, dtype=torch.float32, device=partials.device) if want_stats else None
    _lib.check(_lib.load().fmc_groupnorm_coef(partials.data_ptr(), splits, gamma.data_ptr(), beta.data_ptr(), coef.data_ptr(), _p(stats), N, hw, C,
                                              groups, float(eps), _stream()), "fmc_groupnorm_coef")
    return (coef, stats) if want_stats else coef


def conv3x3_halo4_supported(n: int, h: int, w: int, cin: int, cin1: int, cout: int, upsample: bool, wide: bool = False) -> bool:
    return bool(_lib.load().fmc_conv3x3_halo4_supported(n, h, w, cin, cin1, cout, int(upsample), int(wide)))


def _w_halo4_packed(weight_cl: torch.Tensor, wide: bool = False) -> torch.Tensor:
    cache = _owner_cache(weight_cl, "_fmc_wtm")
    key = ("halo4w" if wide else "halo4", weight_cl.storage_offset(), tuple(weight_cl.shape), tuple(weight_cl.stride()), weight_cl._version)
    hit = cache.get(key)
    if hit is None:
        cout, cin = weight_cl.shape[:2]
        assert weight_cl.is_contiguous(memory_format=torch.channels_last)
        hit = torch.empty(cout * 9 * cin, dtype=weight_cl.dtype, device=weight_cl.device)
        _lib.check(_lib.load().fmc_conv3x3_halo4_pack_weight(weight_cl.data_ptr(), hit.data_ptr(), cin, cout, int(wide), _stream()),
                   "fmc_conv3x3_halo4_pack_weight")
        cache[key] = hit
    return hit


def conv3x3_halo4_split(n: int, h: int, w: int, cin: int, cout: int, cus: int = 256, wide: bool = False) -> int:
    """Workgroups per tile: 1 where the tiles fill the chip, else the divisor of the chunk count that brings the launch closest to one round."""
    tiles = _lib.load().fmc_conv3x3_halo4_tiles(n, h, w, cout, int(wide))
    if tiles >= (3 * cus) // 4:
        return 1
    nchunk = cin // 64
    best = 1
    for s in range(2, min(nchunk, 16) + 1):
        if nchunk % s == 0 and tiles * s <= cus + cus // 8:
            best = s
    return best


def conv3x3_halo4(x_nhwc: torch.Tensor, weight_cl: torch.Tensor, bias=None, temb=None, residual_nhwc=None, temb_div: int = 1, upsample: bool = False,
                  x2_nhwc: Optional[torch.Tensor] = None, emit_gn: bool = False, split_k: Optional[int] = None, wide: bool = False):
    """`conv3x3` on the small feature maps (images 8 / 16 / 32 pixels wide; csrc/conv_halo4.hip): arguments and results as `conv3x3_halo` without the
    GroupNorm operand path; the statistics partials are per (image, row block of 5 / 10 rows)."""
    _dev(x_nhwc, weight_cl, bias, temb, residual_nhwc, x2_nhwc)
    n, hs, ws, c1 = x_nhwc.shape
    h, w = (2 * hs, 2 * ws) if upsample else (hs, ws)
    cout, cin = weight_cl.shape[:2]
    assert x_nhwc.is_contiguous() and x_nhwc.dtype == torch.bfloat16 and weight_cl.dtype == torch.bfloat16
    if x2_nhwc is not None:
        assert x2_nhwc.is_contiguous() and x2_nhwc.shape[:3] == x_nhwc.shape[:3] and x2_nhwc.dtype == x_nhwc.dtype and c1 + x2_nhwc.shape[3] == cin
    else:
        assert c1 == cin
    assert temb is None or (temb.stride(1) == 1 and temb.shape == (n // temb_div, cout) and n % temb_div == 0)
    assert residual_nhwc is None or (residual_nhwc.is_contiguous() and residual_nhwc.shape == (n, h, w, cout))
    L = _lib.load()
    wp = _w_halo4_packed(weight_cl, wide)
    out = torch.empty(n, h, w, cout, dtype=x_nhwc.dtype, device=x_nhwc.device)
    if split_k is None:
        split_k = 1 if emit_gn else conv3x3_halo4_split(n, h, w, cin, cout, wide=wide)
    ws, ws_bytes = (None, 0) if split_k <= 1 else _splitk_workspace(x_nhwc.device, split_k, n * h * w, cout)
    part = torch.empty(n, L.fmc_conv3x3_halo4_row_blocks_per_image(h, w), 32, 2, dtype=torch.float32, device=x_nhwc.device) if emit_gn else None
    conv_halo_calls["conv4"] = conv_halo_calls.get("conv4", 0) + 1
    if call_log is not None:
        call_log.append(("conv_halo4", (n, h, w, cin, cout, bool(upsample)), 2.0 * n * h * w * cout * 9 * cin))
    _lib.check(L.fmc_conv3x3_halo4_bf16(x_nhwc.data_ptr(), _p(x2_nhwc), c1, wp.data_ptr(), _p(bias), _p(temb), _p(residual_nhwc), out.data_ptr(),
                                        n, h, w, cin, cout, 0 if temb is None else temb.stride(0), int(temb_div), int(upsample), _p(part), int(split_k),
                                        ws, ws_bytes, int(wide), _stream()), "fmc_conv3x3_halo4_bf16")
    return (out, part) if emit_gn else out


def groupnorm_partials(x: torch.Tensor, groups: int, x2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The statistics pass of a GroupNorm alone: x `[N, S, C]` (+ `x2 [N, S, C2]`: channel concat read in place) -> partial (sum, sum of squares)
    `[N, splits, G, 2]` (one read of x, no write)."""
    _dev(x, x2)
    N, S, C1 = x.shape
    C = C1 + (x2.shape[2] if x2 is not None else 0)
    assert x.is_contiguous() and (x2 is None or (x2.is_contiguous() and x2.shape[:2] == x.shape[:2] and x2.dtype == x.dtype))
    L = _lib.load()
    part = torch.empty(N, L.fmc_groupnorm_partial_splits(S, C), groups, 2, dtype=torch.float32, device=x.device)
    conv_halo_calls["stats_pass"] += 1
    _lib.check(L.fmc_groupnorm_partials(x.data_ptr(), _p(x2), C1 if x2 is not None else 0, part.data_ptr(), N, S, C, groups, _dt(x), _stream()),
               "fmc_groupnorm_partials")
    return part


def conv3x3_halo(x_nhwc: torch.Tensor, weight_cl: torch.Tensor, bias=None, temb=None, residual_nhwc=None, temb_div: int = 1, upsample: bool = False,
                 x2_nhwc: Optional[torch.Tensor] = None, gn_coef: Optional[torch.Tensor] = None, gn_act: bool = True, emit_gn: bool = False):
    """`conv3x3(act(x * scale + shift))` on channels-last bf16 images: x `[N, Hs, Ws, C1]` (+ `x2 [N, Hs, Ws, C2]`: channel concat read in place),
    filter `[Cout, C1 + C2, 3, 3]` channels-last, `gn_coef [N, C1 + C2, 2]` fp32 or None (plain convolution).  Returns out `[N, H, W, Cout]`, or
    `(out, partials [N, tiles, 32, 2])` with `emit_gn` (statistics of the output for the GroupNorm that consumes it)."""
    _dev(x_nhwc, weight_cl, bias, temb, residual_nhwc, x2_nhwc, gn_coef)
    n, hs, ws, c1 = x_nhwc.shape
    h, w = (2 * hs, 2 * ws) if upsample else (hs, ws)
    cout, cin = weight_cl.shape[:2]
    assert x_nhwc.is_contiguous() and x_nhwc.dtype == torch.bfloat16 and weight_cl.dtype == torch.bfloat16
    if x2_nhwc is not None:
        assert x2_nhwc.is_contiguous() and x2_nhwc.shape[:3] == x_nhwc.shape[:3] and x2_nhwc.dtype == x_nhwc.dtype and c1 + x2_nhwc.shape[3] == cin
    else:
        assert c1 == cin
    assert temb is None or (temb.stride(1) == 1 and temb.shape == (n // temb_div, cout) and n % temb_div == 0)
    assert residual_nhwc is None or (residual_nhwc.is_contiguous() and residual_nhwc.shape == (n, h, w, cout))
    assert gn_coef is None or (gn_coef.shape == (n, cin, 2) and gn_coef.dtype == torch.float32 and gn_coef.is_contiguous())
    L = _lib.load()
    wp = _w_halo_packed(weight_cl)
    out = torch.empty(n, h, w, cout, dtype=x_nhwc.dtype, device=x_nhwc.device)
    part = torch.empty(n, L.fmc_conv3x3_halo_tiles_per_image(h, w), 32, 2, dtype=torch.float32, device=x_nhwc.device) if emit_gn else None
    conv_halo_calls["conv"] += 1
    conv_halo_calls["gn_fused"] += gn_coef is not None
    if call_log is not None:
        call_log.append(("conv_halo", (n, h, w, cin, cout, bool(upsample)), 2.0 * n * h * w * cout * 9 * cin))
    _lib.check(L.fmc_conv3x3_halo_bf16(x_nhwc.data_ptr(), _p(x2_nhwc), c1, wp.data_ptr(), _p(bias), _p(temb), _p(residual_nhwc), out.data_ptr(),
                                       n, h, w, cin, cout, 0 if temb is None else temb.stride(0), int(temb_div), int(upsample), _p(gn_coef),
                                       int(gn_act), _p(part), _stream()), "fmc_conv3x3_halo_bf16")
    return (out, part) if emit_gn else out


# --------------------------------------------------------------------------------------------
# fp32-storage ("parity") mode of the two GEMMs: split-bf16 x3 operands on the same gfx950 kernels, fp32 epilogue
# (include/fmc_hip.h: fmc_split_bf16x3 / fmc_linear_x3_f32 / fmc_conv3x3_x3_f32).  FMC_F32_GEMM=0 sends fp32 projections /
# convolutions back to the vendor libraries (A/B: which part of a parity figure is the product path's own indexing).
# --------------------------------------------------------------------------------------------
F32_GEMM = os.environ.get("FMC_F32_GEMM", "1") != "0"
f32_gemm_calls = {"linear": 0, "geglu": 0, "conv3x3": 0}      # launches through the split-bf16 x3 entry points (tests assert on these)


def split_bf16x3(src: torch.Tensor, role: int, dst: Optional[torch.Tensor] = None, col0: int = 0, K: Optional[int] = None) -> torch.Tensor:
    """fp32 `[..., C]` (dense last dim, uniformly strided rows) -> bf16 `[rows, 3 K]`: activation (role 0) `[hi | hi | lo]`, weight
    (role 1) `[hi | lo | hi]`; `dst` / `col0` / `K` place a second source into the same buffer (two-source operands)."""
    _dev(src)
    assert src.dtype == torch.float32
    C = src.shape[-1]
    rows, ld = _rows2d(src)
    K = C if K is None else K
    if dst is None:
        dst = torch.empty(rows, 3 * K, dtype=torch.bfloat16, device=src.device)
    assert dst.shape == (rows, 3 * K) and dst.is_contiguous()
    _lib.check(_lib.load().fmc_split_bf16x3(src.data_ptr(), dst.data_ptr(), rows, C, ld, 3 * K, col0, K, role, _stream()),
               "fmc_split_bf16x3")
    return dst


def _split_weight_cached(weight: torch.Tensor, rows: int, C: int) -> torch.Tensor:
    """`[hi | lo | hi]` form of a frozen fp32 weight viewed as `[rows, C]`, cached on the tensor per version (views miss: re-split)."""
    hit = getattr(weight, "_fmc_w3", None)
    if hit is None or hit[0] != (weight._version, weight.data_ptr()):
        w2 = weight.detach()
        w2 = w2.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(rows, C) if weight.ndim == 4 else w2.reshape(rows, C)
        hit = ((weight._version, weight.data_ptr()), split_bf16x3(w2.contiguous(), 1))
        try:
            weight._fmc_w3 = hit
        except Exception:
            pass
    return hit[1]


def _f32_arm(key_bf16, tile: int) -> int:
    """Parity mode runs the arm the bf16 product path chose for the same problem when that is a plain-grid arm (so the checked
    code is the timed code); stream-K forms map to their plain geometry, the vendor arm / arm 15 to the kernel's own rule."""
    if tile:
        return tile
    return _f32_arm_of(_choice.get(key_bf16, 0))


def _f32_arm_of(use: int) -> int:
    """bf16 autotune arm id -> the arm parity mode runs.  The 160 x 320 family (512.. / 544.., with or without split-K; 528 = its 256-row GEGLU
    form) maps to ARM_160 -- the same kernel on the row-major split-bf16 weight -- BEFORE the stream-K offsets are peeled (all of them are >= 256);
    128+ / 256+ / 384+ stream-K forms map to their plain geometry."""
    if ARM_160 <= use < ARM_160 + 5 or ARM_160B <= use < ARM_160B + 5 or use == ARM_256:
        return ARM_160
    if use >= 384:
        use -= 384
    elif use >= 256:
        use -= 256
    elif use >= 128:
        use -= 128
    if 16 <= use < 128:                                 # split-K ids (geometry + 16 log2(split)): the geometry
        use &= 15
    return use if 1 <= use <= 14 else 0


def linear_f32(x: torch.Tensor, weight: torch.Tensor, bias=None, residual=None, alpha: float = 1.0, geglu: bool = False,
               tile: int = 0, split_k: int = 1, x2=None, residual2=None) -> torch.Tensor:
    """`linear_bf16` for fp32 tensors: split-bf16 x3 operands, fp32 accumulate, fp32 epilogue (fmc_linear_x3_f32)."""
    tile, split_k = _decode_arm(tile, split_k)
    if split_k < 1:
        split_k = 1
    _dev(x, weight, bias, residual, x2)
    N, Kd = weight.shape
    M, _ = _rows2d(x)
    if x2 is None:
        x3 = split_bf16x3(x, 0)
    else:
        k1 = x.shape[-1]
        assert k1 + x2.shape[-1] == Kd and _rows2d(x2)[0] == M
        x3 = torch.empty(M, 3 * Kd, dtype=torch.bfloat16, device=x.device)
        split_bf16x3(x, 0, x3, 0, Kd)
        split_bf16x3(x2, 0, x3, k1, Kd)
    w3 = _split_weight_cached(weight, N, Kd)
    n_out = N // 2 if geglu else N
    out = torch.empty(*x.shape[:-1], n_out, dtype=torch.float32, device=x.device)
    ldres = 0
    if residual is not None:
        assert residual.shape == out.shape and residual.dtype == torch.float32
        _, ldres = _rows2d(residual)
    if residual2 is not None:
        assert residual is not None and residual2.shape == out.shape and _rows2d(residual2)[1] == ldres
    b = None if bias is None else bias.float()
    ws, ws_bytes = _splitk_workspace(x.device, split_k, M, N)
    f32_gemm_calls["geglu" if geglu else "linear"] += 1
    _lib.check(_lib.load().fmc_linear_x3_f32(x3.data_ptr(), w3.data_ptr(), _p(b), _p(residual), out.data_ptr(), M, N, 3 * Kd,
                                             3 * Kd, ldres, n_out, float(alpha), int(geglu), int(tile), int(split_k), ws, ws_bytes,
                                             _p(residual2), _stream()), "fmc_linear_x3_f32")
    return out


def conv3x3_f32(x_nhwc: torch.Tensor, weight: torch.Tensor, bias=None, temb=None, residual_nhwc=None, tile: int = 0,
                split_k: int = 1, temb_div: int = 1, upsample: bool = False, stride2: bool = False) -> torch.Tensor:
    """`conv3x3_bf16` for fp32 tensors (x `[N, H, W, Cin]` contiguous, weight logical `[Cout, Cin, 3, 3]`)."""
    tile, split_k = _decode_arm(tile, split_k)
    if split_k < 1:
        split_k = 1
    _dev(x_nhwc, weight, bias, temb, residual_nhwc)
    n, h, w, cin = x_nhwc.shape
    if upsample:
        h, w = 2 * h, 2 * w
    if stride2:
        assert not upsample and h % 2 == 0 and w % 2 == 0
        h, w = h // 2, w // 2
    cout = weight.shape[0]
    assert x_nhwc.is_contiguous() and x_nhwc.dtype == torch.float32
    assert temb is None or (temb.stride(1) == 1 and temb.shape == (n // temb_div, cout) and temb.dtype == torch.float32)
    assert residual_nhwc is None or (residual_nhwc.is_contiguous() and residual_nhwc.shape == (n, h, w, cout))
    x3 = split_bf16x3(x_nhwc.view(-1, cin), 0)
    w3 = _split_weight_cached(weight, cout * 9, cin)
    out = torch.empty(n, h, w, cout, dtype=torch.float32, device=x_nhwc.device)
    b = None if bias is None else bias.float()
    ws, ws_bytes = _splitk_workspace(x_nhwc.device, split_k, n * h * w, cout)
    f32_gemm_calls["conv3x3"] += 1
    _lib.check(_lib.load().fmc_conv3x3_x3_f32(x3.data_ptr(), w3.data_ptr(), _p(b), _p(temb), _p(residual_nhwc), out.data_ptr(),
                                              n, h, w, 3 * cin, cout, 0 if temb is None else temb.stride(0), int(temb_div),
                                              2 if stride2 else int(upsample), int(tile), int(split_k), ws, ws_bytes, _stream()),
               "fmc_conv3x3_x3_f32")
    return out


# --------------------------------------------------------------------------------------------
# projection / convolution front-ends: pick, per problem shape, between the fused gfx950 kernel and the vendor
# library call (+ separate epilogue passes).  The choice is measured once per shape on the first eager call (the
# pipelines run eager warm-up steps before capturing a HIP graph) and cached; while a graph is being captured an
# unseen shape falls back to a static rule.  Both arms compute the same function.
# --------------------------------------------------------------------------------------------
_choice = {}
_tune_log = {}      # key -> {arm: ms} measured when the choice was made
_calls = {}         # key -> eager calls seen (graph replays do not pass through Python)
# FMC_AUTOTUNE=0: never time anything -- unseen shapes take the static rule (reproducible arm choice for tests / goldens:
# split-K and stream-K arms change the summation order, so a timing-dependent choice changes low-order bits run to run)
AUTOTUNE = os.environ.get("FMC_AUTOTUNE", "1") != "0"
# FMC_DETERMINISTIC=1: the vendor convolution arm is never chosen (MIOpen's small-image 3x3 kernels accumulate with atomics:
# bit-different results run to run, measured on the 4x4 level of the camera encoder); every fmc_* kernel is deterministic
DETERMINISTIC = os.environ.get("FMC_DETERMINISTIC", "0") == "1"
# The measured table persists on disk, keyed by the sha256 of the library build and the device name: a second process
# (the next sampling run, every rank of a multi-GPU job) starts tuned and picks the SAME arms.  FMC_AUTOTUNE_CACHE=path
# moves the file, FMC_AUTOTUNE_CACHE=0 disables persistence.
_CACHE_ENV = os.environ.get("FMC_AUTOTUNE_CACHE", "")
_cache_state = {"loaded": False, "dirty": False, "meta": None}


def _cache_path():
    if _CACHE_ENV == "0":
        return None
    return _CACHE_ENV or os.path.join(os.path.dirname(_lib.LIB_PATH), "autotune_cache.json")


def _cache_meta():
    if _cache_state["meta"] is None:
        import hashlib
        with open(_lib.LIB_PATH, "rb") as f:
            sha = hashlib.sha256(f.read()).hexdigest()[:16]
        _cache_state["meta"] = {"lib_sha16": sha, "device": torch.cuda.get_device_name() if torch.cuda.is_available() else "cpu",
                                "arms": list(GEMM_TILES), "hipblaslt": vendor_version() if torch.cuda.is_available() else 0}
    return _cache_state["meta"]


DEFAULT_ARM_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "autotune_default_mi355x.json")
stale_table_entries = [0]                          # table entries dropped by `_pick` because their arm is disabled / ineligible in this process
autotune_sources = {"cache": 0, "defaults": 0}     # shapes taken from the per-build cache / from the tracked default table (bench.py reports them)


def _load_table_file(path: str, strict_meta: bool) -> int:
    import ast
    import json
    if not path or not os.path.isfile(path):
        return 0
    try:
        with open(path) as f:
            blob = json.load(f)
        meta, mine = blob.get("meta") or {}, _cache_meta()
        if strict_meta and meta != mine:
            return 0
        if not strict_meta:                        # the tracked defaults: made for this ARCHITECTURE (device names differ from box to box)
            arch = torch.cuda.get_device_properties(torch.cuda.current_device()).gcnArchName.split(":")[0]
            if meta.get("arch") != arch:
                return 0
        n = 0
        for k, v in blob["choices"].items():
            key = ast.literal_eval(k)
            if key[0] == "valgo" and meta.get("hipblaslt") != mine["hipblaslt"]:
                continue                           # an index into ANOTHER hipBLASLt version's candidate list names other kernels: re-time on this one
            if key not in _choice:
                _choice[key] = int(v["arm"])
                _tune_log[key] = {int(a): ms for a, ms in v.get("ms", {}).items()}
                n += 1
        return n
    except Exception:                              # a corrupt table must never take the run down: re-tune
        return 0


def load_autotune_table(path: Optional[str] = None) -> int:
    """Read the persisted arm tables.  First the per-build cache next to the library (ignored unless it was made by THIS library build on this
    device type and with this arm list); then, for shapes it does not hold, the TRACKED default table `autotune_default_mi355x.json` (made on an
    MI355X by `tools/make_default_arm_table.py` from a bench run; device type must match): a fresh box then runs the arms the table was
    measured with instead of letting timing noise pick them anew (arm tables differed box to box).  `FMC_AUTOTUNE_DEFAULTS=0`
    ignores the tracked table.  Returns the number of shapes loaded.  Called lazily by the first front-end call."""
    _cache_state["loaded"] = True
    n = _load_table_file(path or _cache_path(), True)
    autotune_sources["cache"] += n
    if os.environ.get("FMC_AUTOTUNE_DEFAULTS", "1") != "0" and torch.cuda.is_available():
        d = _load_table_file(DEFAULT_ARM_TABLE, False)
        autotune_sources["defaults"] += d
        n += d
    return n


def save_autotune_table(path: Optional[str] = None) -> Optional[str]:
    """Write the arm table next to the library (atomic rename); rank 0 / single process only is the caller's business."""
    import json
    path = path or _cache_path()
    if not path or not _choice:
        return None
    blob = {"meta": _cache_meta(),
            "choices": {repr(k): {"arm": v, "ms": _tune_log.get(k, {})} for k, v in _choice.items()}}
    tmp = f"{path}.{os.getpid()}.tmp"
    try:
        with open(tmp, "w") as f:
            json.dump(blob, f)
        os.replace(tmp, path)
    except OSError:
        return None
    _cache_state["dirty"] = False
    return path


def _save_at_exit():
    if _cache_state["dirty"] and os.environ.get("RANK", "0") == "0":
        save_autotune_table()


atexit.register(_save_at_exit)
GEMM_TILES = (1, 2, 3, 4, 5, 6, 7, 11,     # 8..10, 12 (4-stage rings) exist but never won on the FMC shapes
              13,                            # the 8-phase 256x256 kernel (staggered wave rows, half-tile DMA, counted vmcnt)
              15,                            # K = 320 token projections: persistent, weights resident in registers (falls back to 5 elsewhere)
              ARM_160B if W_TILEMAJOR else ARM_160,   # 160 x 320 tiles (whole rounds / no padded columns for N = 320 k; falls back to 13 elsewhere), reading the weight
                                             # pre-packed tile-major (bit-identical to ARM_160, -1 .. -5 % per launch: tools/scratch/probe_wtm.py)
              128 + 2, 128 + 3,              # stream-K (persistent workgroups) on the two 1-per-CU geometries
              128 + 13,                      # stream-K on the 8-phase kernel: persistent partial pass + one finishing workgroup per tile
              256 + 13,                      # the same for the LAST PARTIAL ROUND of tiles only, the whole rounds on the plain grid
              384 + 13)                      # k-lockstep split on the 8-phase kernel (round 4): the filter is read once per XCD, lean finishing pass
if os.environ.get("FMC_GEMM_ARMS"):              # A/B switch: the arm list the autotuner may choose from, e.g. "1,2,3,11"
    GEMM_TILES = tuple(int(a) for a in os.environ["FMC_GEMM_ARMS"].split(","))
# fmc_linear_bf16 / fmc_conv3x3_bf16 `tile` arms tried per shape (14 = arm 13 with a deeper prefetch measured no better anywhere,
# tools/scratch/probe_g8.py)


def autotune_report():
    """[(key, chosen arm, {arm: ms}, eager calls)] for every shape tuned so far (arm 0 = vendor library)."""
    return [(k, _choice[k], _tune_log.get(k, {}), _calls.get(k, 0)) for k in _choice]


_tune_stream = None
def _graph_ms(body, reps):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=_tune_stream):
        for _ in range(reps):
            body()
    g.replay()
    _tune_stream.synchronize()
    ms = float("inf")
    for _ in range(3):                              # min of 3: one noisy sample must not pick the arm
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        _tune_stream.synchronize()
        ms = min(ms, e0.elapsed_time(e1) / reps)
    del g
    return ms


def _time_ms(fn, reps=8):
    """GPU time of one `fn()` call, measured on a captured HIP graph of `reps` calls so that the Python / launch
    overhead of the 10-30 us kernels does not decide the arm (in the pipelines they replay from a graph as well)."""
    global _tune_stream
    if _tune_stream is None:
        _tune_stream = torch.cuda.Stream()
    cur = torch.cuda.current_stream()
    _tune_stream.wait_stream(cur)
    with torch.cuda.stream(_tune_stream):
        fn()                                            # lazy initialisation (workspaces, library heuristics) outside capture
        _tune_stream.synchronize()
        ms = _graph_ms(fn, reps)
    cur.wait_stream(_tune_stream)
    return ms


NO_VENDOR = os.environ.get("FMC_NO_VENDOR", "0") == "1"      # A/B switch: the vendor library (hipBLASLt / MIOpen) is never a candidate arm
# where the calls of the three GEMM-shaped front-ends went (bench.py counts one eager step): "own" = a kernel of this library, "vendor" = the
# autotuner chose the vendor arm, "ineligible" = the shape / dtype / layout is outside the own kernels and the call fell through to the library
dispatch_calls = {k: {"own": 0, "vendor": 0, "ineligible": 0} for k in ("linear", "geglu", "conv3x3")}
call_log = None                     # a list while bench.py records one eager step: (front-end, shape tuple, algorithmic flops) per launch


def _log_call(family: str, shape: tuple, flops: float) -> None:
    """One launch of a GEMM-shaped front-end into `call_log` (bench.py: flops per kernel family of the step, and -- by launch order within a
    family -- the in-step duration of one SHAPE of a kernel that serves several).  Families: "own_linear" (one gemm* launch of this library),
    "vendor" (one hipBLASLt launch), "geglu_direct", "fused_block" (temporal / text cross-attention block), "conv_halo", "conv_halo4"."""
    if call_log is not None:
        call_log.append((family, tuple(shape), float(flops)))


def _pick(key, hip_fn, lib_fn, static_hip: bool, extra_arms=(), k320: bool = False, own_only: bool = False) -> int:
    """0 = vendor library arm, 1..6 = fused gfx950 kernel with that tile geometry (`hip_fn(tile)`)."""
    if not _cache_state["loaded"]:
        load_autotune_table()
    no_lib = (DETERMINISTIC and key[0] == "conv") or own_only or NO_VENDOR
    if no_lib:
        key = key + ("det",)
    use = _choice.get(key)
    _calls[key] = _calls.get(key, 0) + 1
    n320 = (key[5] if key[0] == "conv" else key[2]) % 320 == 0
    # the arms THIS call may run: the enabled tile list (FMC_GEMM_ARMS / W_TILEMAJOR) + the caller's gated extras (split arms, small-M tiles, arm 700 only
    # where LINEAR4 and its shape gate hold), minus the forms that exist for some shapes only
    cands = tuple(t for t in GEMM_TILES + tuple(extra_arms)
                  if (t != 15 or k320)    # (arm 15 exists for the K = 320 token projections only,
                  and (not (ARM_160 <= t < ARM_160 + 5 or ARM_160B <= t < ARM_160B + 5 or t == ARM_256) or n320))  #  arm 16 for outputs whose width is a multiple of 320)
    if use is not None and not (use in cands or (use == 0 and not no_lib) or (use == -1 and key not in _tune_log)):
        # a table entry (tracked defaults, an older cache) naming an arm that is switched off or not eligible here: an A/B run must measure the arms it
        # asked for, and a gated arm (700 on a strided x, 15 off its shape) must not be reached through the table -- drop the entry, re-tune below
        _choice.pop(key, None)
        _tune_log.pop(key, None)
        stale_table_entries[0] += 1
        use = None
    if use is None:
        if not AUTOTUNE or torch.cuda.is_current_stream_capturing():
            return 0 if not (static_hip or no_lib) else -1          # -1: kernel's own geometry heuristic
        times = ([] if no_lib else [(_time_ms(lib_fn), 0)]) + [(_time_ms(lambda t=t: hip_fn(t)), t) for t in cands]
        use = min(times)[1]
        _choice[key] = use
        _tune_log[key] = {arm: round(ms, 4) for ms, arm in times}
        _cache_state["dirty"] = True
    return use


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
           residual: Optional[torch.Tensor] = None, alpha: float = 1.0, x2: Optional[torch.Tensor] = None,
           residual2: Optional[torch.Tensor] = None, gn_hw: int = 0, ln: Optional["LnSpec"] = None,
           lazy_residual: bool = False) -> torch.Tensor:
    """`alpha * (x @ weight^T + bias) + residual [+ residual2]` for bf16 device tensors (see `linear_bf16`).  With `x2`
    the input is the concat `[x, x2]` along the last dim; the fused kernel reads the two tensors in place."""
    import torch.nn.functional as F
    pend = getattr(x, "_fmc_pending_ln", None)
    if pend is not None:                                # x's LayerNorm is still to be applied (LayerNorm.forward(defer=True))
        if (x2 is None and residual is None and residual2 is None and alpha == 1.0 and not gn_hw and ln is None and weight.is_contiguous()
                and lnc_ok(x, weight)):
            dispatch_calls["linear"]["own"] += 1
            return linear_lnc(x, weight, bias, pend)
        x = resolve_pending_ln(x)

    def lib():
        xin = x if x2 is None else torch.cat([x, x2], dim=-1)
        if alpha == 1.0 and vendor_linear_ok(xin, weight, bias, residual):
            y = vendor_linear(xin, weight, bias, residual)         # bias + residual inside the library's epilogue: no elementwise pass behind the GEMM
            return y if residual2 is None else y + residual2
        y = F.linear(xin, weight, bias)
        _log_call("vendor", (xin.numel() // xin.shape[-1], weight.shape[0], weight.shape[1], bias is not None, False), 2.0 * xin.numel() * weight.shape[0])
        if residual is not None:
            y = torch.add(residual, y, alpha=alpha)
            return y if residual2 is None else y + residual2
        return y if alpha == 1.0 else y * alpha

    N, Kd = weight.shape
    if (F32_GEMM and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and Kd % 64 == 0 and N % 8 == 0
            and x.shape[-1] % 8 == 0 and (x.is_contiguous() or x.ndim == 2) and (x2 is None or x2.is_contiguous())
            and (residual is None or residual.dtype == torch.float32)):
        M = x.numel() // x.shape[-1]
        key = ("lin", M, N, Kd, bias is not None, (residual is not None) + (residual2 is not None), 0 if x2 is None else x.shape[-1])
        return linear_f32(x, weight, bias, residual, alpha, tile=_f32_arm(key, 0), x2=x2, residual2=residual2)
    ok = x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and Kd % 64 == 0 and N % 8 == 0
    if x2 is None:
        ok = ok and linear_supported(x, weight)
    else:
        ok = ok and x.shape[-1] % 64 == 0 and x2.dtype == x.dtype and x2.is_contiguous()
    if not ok or (x.ndim > 2 and not x.is_contiguous()):
        dispatch_calls["linear"]["ineligible"] += 1
        return lib()
    M = x.numel() // x.shape[-1]
    if ln is not None and x2 is None and not gn_hw and ln_emit_ok(x, weight, residual, residual2, ln):
        # the consumer is a LayerNorm and the 160 x 320 tile holds whole rows (N == 320): it leaves the epilogue too, x is not read again
        dispatch_calls["linear"]["own"] += 1
        return linear_ln(x, weight, bias, residual, alpha, residual2, ln)
    if (gn_hw and x2 is None and x.is_contiguous() and gn_emit_ok(M, N, Kd, gn_hw, x.dtype)
            and (residual is None or (residual.is_contiguous() and residual.dtype == x.dtype))
            and (residual2 is None or residual2.is_contiguous())):
        # the consumer is a GroupNorm at a level where it would read x twice: the 160 x 320 kernel emits the statistics from its epilogue
        dispatch_calls["linear"]["own"] += 1
        return linear_gn(x, weight, bias, residual, alpha, residual2, gn_hw)
    key = ("lin", M, N, Kd, bias is not None, (residual is not None) + (residual2 is not None),
           0 if x2 is None else x.shape[-1])
    hip = lambda tile: (linear4_bf16(x, weight, bias, residual, alpha, residual2) if tile == ARM_G4
                        else linear_bf16(x, weight, bias, residual, alpha, tile=tile, x2=x2, residual2=residual2))
    small = (ARM_SMALLM, ARM_SMALLM + 2) if (M <= 2560 and x2 is None and residual2 is None) else ()      # 64 x 128 tiles: 200-400 workgroups where 128 x 128 gives 100-200
    # the software-pipelined 160 x 160 kernel: a candidate wherever its tiles are at most two rounds of the chip (the inner levels' projections)
    t160 = ((M + 159) // 160) * ((N + 159) // 160)
    if (LINEAR4 and x2 is None and weight.is_contiguous() and 48 <= t160 <= 640 and Kd >= 320
            and _lib.load().fmc_linear4_supported(M, N, Kd, _rows2d(x)[1])                       # (a strided x with ldx % 8 != 0 must not reach the tuner's timing loop)
            and (residual is None or _rows2d(residual)[1] % 8 == 0)
            and (residual2 is None or (residual is not None and _rows2d(residual2)[1] == _rows2d(residual)[1]))):
        small = small + (ARM_G4,)
    use = _pick(key, hip, lib, Kd <= 640 and N <= 1024 and M >= 16384, split_arms(M, N, Kd) + small,
                k320=(Kd == 320 and N % 320 == 0 and M % 64 == 0 and x2 is None and residual2 is None))
    if (use == 0 and lazy_residual and LAZY_RESIDUAL and residual is not None and residual2 is None and alpha == 1.0 and x2 is None
            and N in (320, 640, 1280) and residual.is_contiguous() and not torch.is_grad_enabled()
            and not vendor_linear_ok(x, weight, bias, residual)):
        # vendor arm through torch (FMC_VENDOR_DIRECT=0): the caller's next op is a LayerNorm of `y + residual` (LayerNorm.skip): it does the add in its
        # own pass.  With the direct hipBLASLt call the residual rides in the GEMM's epilogue instead and the LayerNorm reads one tensor (-0.08 ms, same-box A/B)
        dispatch_calls["linear"]["vendor"] += 1
        y = F.linear(x, weight, bias)
        y._fmc_pending_add = residual
        return y
    dispatch_calls["linear"]["vendor" if use == 0 else "own"] += 1
    return lib() if use == 0 else hip(max(use, 0))


def geglu_linear(x: torch.Tensor, weight: torch.Tensor, bias, weight_il: torch.Tensor, bias_il, weight_il160=None,
                 bias_il160=None) -> torch.Tensor:
    """GEGLU feed-forward input projection: `a * gelu(g)`, `a, g = (x @ weight^T + bias).chunk(2)`.  `weight_il` /
    `bias_il` are the tile-interleaved copies the fused kernel wants (`models.layers.interleave_geglu`); `weight_il160` /
    `bias_il160` the [8 value | 8 gate]-per-16 order of the 160 x 320 arm (without them arm 16 is not a candidate)."""
    import torch.nn.functional as F
    pend = getattr(x, "_fmc_pending_ln", None)
    if pend is not None:
        if weight_il160 is not None and weight_il160.is_contiguous() and lnc_ok(x, weight_il160):
            return linear_lnc(x, weight_il160, bias_il160, pend, geglu=True)
        x = resolve_pending_ln(x)
    lib = lambda: geglu(F.linear(x, weight, bias))
    if (F32_GEMM and x.is_cuda and x.dtype == torch.float32 and weight_il.dtype == torch.float32 and weight_il.shape[0] % 64 == 0
            and weight_il.shape[1] % 64 == 0 and x.is_contiguous()):
        N, Kd = weight.shape
        arm = _f32_arm(("geglu", x.numel() // Kd, N, Kd), 0)
        if arm in (ARM_160, ARM_160B) and weight_il160 is not None:
            return linear_f32(x, weight_il160, bias_il160, geglu=True, tile=ARM_160)
        return linear_f32(x, weight_il, bias_il, geglu=True, tile=0 if arm in (ARM_160, ARM_160B) else arm)
    if not linear_supported(x, weight_il) or weight_il.shape[0] % 64 or (x.ndim > 2 and not x.is_contiguous()):
        dispatch_calls["geglu"]["ineligible"] += 1
        return lib()
    N, Kd = weight.shape
    M = x.numel() // Kd
    has160 = weight_il160 is not None and N % 320 == 0

    def hip(tile):
        if tile in (ARM_160, ARM_256, ARM_160B):        # (without the 160-block order these arms would silently pair wrong rows: route them to 13)
            return linear_bf16(x, weight_il160, bias_il160, geglu=True, tile=tile) if has160 else linear_bf16(x, weight_il, bias_il, geglu=True, tile=13)
        return linear_bf16(x, weight_il, bias_il, geglu=True, tile=tile)
    # (ARM_256, the persistent 256 x 320 form, is selectable but not a candidate: measured 235 / 153 / 134 us against ARM_160's 203 / 153 / 130 us on
    #  the three U-Net levels, tools/scratch/probe_g256.py -- DESIGN.md section 6, round 3.  FMC_GEMM_ARMS=...,528 offers it.)
    use = _pick(("geglu", M, N, Kd), hip, lib, M >= 65536)
    dispatch_calls["geglu"]["vendor" if use == 0 else "own"] += 1
    return lib() if use == 0 else hip(max(use, 0))


def conv3x3(x_nchw: torch.Tensor, weight_cl: torch.Tensor, bias, temb=None, residual_nchw=None, stride=(1, 1),
            padding=(1, 1), temb_div: int = 1, upsample: bool = False, emit_gn: bool = False, own_only: bool = False) -> torch.Tensor:
    """3x3 conv on a logical NCHW / physical channels-last tensor with `+ temb[:, :, None, None]` and `+ residual`.
    Returns a logical NCHW view over channels-last storage.  `own_only`: the vendor library is not a candidate arm."""
    import torch.nn.functional as F

    def lib():
        xin = F.interpolate(x_nchw, scale_factor=2.0, mode="nearest") if upsample else x_nchw
        y = F.conv2d(xin, weight_cl, bias, stride, padding)
        if temb is not None:
            y = y + (temb if temb_div == 1 else temb.repeat_interleave(temb_div, dim=0))[:, :, None, None]
        if residual_nchw is not None:
            y = y + residual_nchw
        return y

    if (F32_GEMM and x_nchw.is_cuda and x_nchw.dtype == torch.float32 and weight_cl.dtype == torch.float32
            and tuple(weight_cl.shape[2:]) == (3, 3) and tuple(padding) == (1, 1) and weight_cl.shape[1] % 64 == 0
            and weight_cl.shape[0] % 8 == 0 and (tuple(stride) == (1, 1) or (tuple(stride) == (2, 2) and not upsample
                                                                           and x_nchw.shape[-1] % 2 == 0 and x_nchw.shape[-2] % 2 == 0))):
        n, cin, h, w = x_nchw.shape
        x = x_nchw.permute(0, 2, 3, 1)
        r = None if residual_nchw is None else residual_nchw.permute(0, 2, 3, 1)
        if x.is_contiguous() and (r is None or r.is_contiguous()) and (temb is None or temb.stride(1) == 1):
            s2 = tuple(stride) == (2, 2)
            ho, wo = (2 * h, 2 * w) if upsample else ((h // 2, w // 2) if s2 else (h, w))
            key = ("conv", n, ho, wo, cin, weight_cl.shape[0], temb is not None, r is not None, upsample, s2)
            return conv3x3_f32(x, weight_cl, bias, temb, r, tile=_f32_arm(key, 0), temb_div=temb_div, upsample=upsample,
                               stride2=s2).permute(0, 3, 1, 2)
    if not conv3x3_supported(x_nchw, weight_cl, stride, padding):
        dispatch_calls["conv3x3"]["ineligible"] += 1
        return lib()
    n, cin, h, w = x_nchw.shape
    cout = weight_cl.shape[0]
    x = x_nchw.permute(0, 2, 3, 1)
    r = None if residual_nchw is None else residual_nchw.permute(0, 2, 3, 1)
    if not x.is_contiguous() or (r is not None and not r.is_contiguous()):
        dispatch_calls["conv3x3"]["ineligible"] += 1
        return lib()
    if upsample:
        h, w = 2 * h, 2 * w
    stride2 = tuple(stride) == (2, 2)
    if stride2:
        h, w = h // 2, w // 2
    # the halo-resident kernel (csrc/conv_halo.hip): 1.75 - 1.95 x the ring kernels wherever its 10 x 32 pixel x 160 channel tiles fill the chip
    # (tools/scratch/r05/bench_halo.py); every such convolution also leaves the statistics of the GroupNorm that consumes its output
    if (CONV_HALO and not stride2 and (temb is None or temb.stride(1) == 1) and conv3x3_halo_supported(n, h, w, cin, cin, cout, upsample)
            and n * _lib.load().fmc_conv3x3_halo_tiles_per_image(h, w) * (cout // 160) >= CONV_HALO_MIN_TILES):
        want = bool(emit_gn and GN_EPILOGUE and cout % 64 == 0 and 160 % (cout // 32) == 0 and not torch.is_grad_enabled())
        dispatch_calls["conv3x3"]["own"] += 1
        y = conv3x3_halo(x, weight_cl, bias, temb, r, temb_div, upsample, emit_gn=want)
        if want:
            y, part = y
            gn_epilogue_calls["emitted"] += 1
            y = y.permute(0, 3, 1, 2)
            y._fmc_gn = (part, cout)
            return y
        return y.permute(0, 3, 1, 2)
    # ... and its 4-wave form for the small feature maps (images 8 / 16 pixels wide: the two inner levels; csrc/conv_halo4.hip), split over the
    # reduction where the tiles alone would leave most of the chip idle (5x8-pixel images: 64 tiles)
    if (CONV_HALO and CONV_HALO4 and not stride2 and w % 8 == 0 and (temb is None or temb.stride(1) == 1)
            and conv3x3_halo4_supported(n, h, w, cin, cin, cout, upsample)
            and _lib.load().fmc_conv3x3_halo4_tiles(n, h, w, cout, 0) * conv3x3_halo4_split(n, h, w, cin, cout) >= CONV_HALO_MIN_TILES):
        dispatch_calls["conv3x3"]["own"] += 1
        # (statistics for the consuming GroupNorm where it would otherwise read its input twice -- the two-pass norm of the large feature maps -- and
        #  the launch is not split over the reduction)
        want = bool(emit_gn and GN_EPILOGUE and h * w >= GN_MIN_HW and cout % 64 == 0 and 80 % (cout // 32) == 0 and not torch.is_grad_enabled()
                    and conv3x3_halo4_split(n, h, w, cin, cout) == 1)
        y = conv3x3_halo4(x, weight_cl, bias, temb, r, temb_div, upsample, emit_gn=want)
        if want:
            y, part = y
            gn_epilogue_calls["emitted"] += 1
            y = y.permute(0, 3, 1, 2)
            y._fmc_gn = (part, cout)
            return y
        return y.permute(0, 3, 1, 2)
    if emit_gn and gn_emit_ok(n * h * w, cout, 9 * cin, h * w, x.dtype):
        dispatch_calls["conv3x3"]["own"] += 1
        y, tag = conv3x3_gn(x, weight_cl, bias, temb, r, temb_div, upsample, stride2)
        y = y.permute(0, 3, 1, 2)
        y._fmc_gn = tag
        return y
    key = ("conv", n, h, w, cin, cout, temb is not None, r is not None, upsample, stride2)
    hip = lambda tile: conv3x3_bf16(x, weight_cl, bias, temb, r, tile=tile, temb_div=temb_div,
                                    upsample=upsample, stride2=stride2).permute(0, 3, 1, 2)
    tiles = ((n * h * w + 127) // 128) * ((cout + 127) // 128)
    use = _pick(key, hip, lib, tiles >= 256, split_arms(n * h * w, cout, 9 * cin), own_only=own_only)
    dispatch_calls["conv3x3"]["vendor" if use == 0 else "own"] += 1
    return lib() if use == 0 else hip(max(use, 0))


# --------------------------------------------------------------------------------------------
# frozen-weight autograd wrappers (training stages 2-3: the U-Net is frozen, only the activation gradient flows through
# it, train_cam_obj_ctrl.py:242).  Forward = the fused front-ends above; backward-data = the same kernels:
#   conv3x3:  dX = conv3x3(dY, W') with W'[ci, ky, kx, co] = W[co, 2-ky, 2-kx, ci]   (pad 1, stride 1)
#   linear:   dX = alpha * dY @ W,  d(residual) = dY
# --------------------------------------------------------------------------------------------
def _flipped_filter(weight_cl: torch.Tensor) -> torch.Tensor:
    """`[Cin, Cout, 3, 3]` filter of the backward-data convolution, channels-last memory format.  Cached ON the weight
    tensor (an attribute, keyed by its version counter): a global dict keyed by `data_ptr()` hands out stale filters
    when a freed weight's address is reused by another model."""
    hit = getattr(weight_cl, "_fmc_flipped", None)
    if hit is None or hit[0] != weight_cl._version:
        hit = (weight_cl._version,
               weight_cl.detach().flip(2, 3).permute(1, 0, 2, 3).contiguous(memory_format=torch.channels_last))
        weight_cl._fmc_flipped = hit
    return hit[1]


class _Conv3x3Frozen(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight_cl, bias, temb, residual, temb_div):
        ctx.save_for_backward(weight_cl)
        ctx.has_res = residual is not None and residual.requires_grad
        with torch.no_grad():
            return conv3x3(x, weight_cl, bias, temb, residual, (1, 1), (1, 1), temb_div)

    @staticmethod
    def backward(ctx, dy):
        (weight_cl,) = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = None
        if ctx.needs_input_grad[0]:
            with torch.no_grad():
                dx = conv3x3(dy, _flipped_filter(weight_cl), None)
        return dx, None, None, None, (dy if ctx.has_res else None), None


def conv3x3_frozen(x, weight_cl, bias, temb=None, residual=None, temb_div: int = 1):
    """3x3 / stride 1 / pad 1 conv with frozen filter, differentiable w.r.t. x (and the residual)."""
    return _Conv3x3Frozen.apply(x, weight_cl, bias, temb, residual, temb_div)


def _transposed_weight(weight: torch.Tensor) -> torch.Tensor:
    """`W^T [K, N]` contiguous: the backward-data GEMM `dX = dY @ W` is `fmc_linear_bf16(dY, W^T)` -- both operands
    reduction-contiguous.  Cached ON THE TENSOR THAT OWNS THE STORAGE (`weight._base` for a view, else the weight itself), keyed by
    (offset, shape, strides, version): frozen weights reached through a fresh view every call (`weight.view(out, in)` of a 1x1 conv) hit
    the cache, and the entry dies with its owner.  (A module-level dict keyed by the storage pointer -- the first form of this cache --
    handed a freed weight's W^T to the next model whose weight landed on the same address with the same shape: wrong gradients,
    silently.)"""
    owner = weight._base if weight._base is not None else weight
    key = (weight.storage_offset(), tuple(weight.shape), tuple(weight.stride()), weight._version, weight.dtype)
    cache = getattr(owner, "_fmc_wt", None)
    if cache is None or cache[0] != owner._version:
        cache = (owner._version, {})
        try:
            owner._fmc_wt = cache
        except Exception:                                   # (an owner that takes no attributes: transpose per call)
            pass
    hit = cache[1].get(key)
    if hit is None:
        hit = weight.detach().t().contiguous()
        cache[1][key] = hit
    return hit


def linear_backward_data(dy: torch.Tensor, weight: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """`alpha * dY @ W` for a frozen `W [N, K]` on the fused gfx950 GEMM (autotuned like every projection)."""
    if dy.is_cuda and dy.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and weight.shape[0] % 64 == 0 \
            and weight.shape[1] % 8 == 0:
        with torch.no_grad():
            return linear(dy if dy.is_contiguous() else dy.contiguous(), _transposed_weight(weight), None, None, alpha)
    dx = torch.matmul(dy, weight)
    return dx if alpha == 1.0 else dx * alpha


class _LinearFrozen(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, alpha):
        ctx.save_for_backward(weight)
        ctx.alpha = alpha
        ctx.has_res = residual is not None and residual.requires_grad
        with torch.no_grad():
            return linear(x, weight, bias, residual, alpha)

    @staticmethod
    def backward(ctx, dy):
        (weight,) = ctx.saved_tensors
        dx = linear_backward_data(dy, weight, ctx.alpha) if ctx.needs_input_grad[0] else None
        return dx, None, None, (dy if ctx.has_res else None), None


def linear_frozen(x, weight, bias=None, residual=None, alpha: float = 1.0):
    """`alpha * (x @ W^T + b) + residual` with frozen W, b; differentiable w.r.t. x and the residual."""
    return _LinearFrozen.apply(x, weight, bias, residual, alpha)


def conv3x3_weight_grad(x_nhwc: torch.Tensor, dy_nhwc: torch.Tensor) -> torch.Tensor:
    """dW `[Cout, Cin, 3, 3]` (bf16, a permuted view) of a 3x3 / stride 1 / pad 1 convolution from its NHWC input and output
    gradient: two `fmc_nhwc_to_cmajor_padded` layout passes + one split-K `fmc_linear_bf16` per kernel row (see fmc_hip.h)."""
    _dev(x_nhwc, dy_nhwc)
    n, H, W, cin = x_nhwc.shape
    cout = dy_nhwc.shape[-1]
    assert dy_nhwc.shape[:3] == x_nhwc.shape[:3] and x_nhwc.is_contiguous() and dy_nhwc.is_contiguous()
    assert x_nhwc.dtype == torch.bfloat16 and dy_nhwc.dtype == torch.bfloat16 and cin % 8 == 0 and cout % 8 == 0
    Hp, Wp = H + 2, (W + 2 + 7) // 8 * 8
    Lk = (n * Hp * Wp + 63) // 64 * 64                 # reduction length (padded pixels), a multiple of the k-tile
    G = Wp + 8                                          # guard in front of / behind every X^T row: a tap reaches +-Wp
    Lg = Lk + 2 * G
    lib = _lib.load()
    xt = torch.empty(3, cin, Lg, dtype=torch.bfloat16, device=x_nhwc.device)
    dyt = torch.empty(cout, Lk, dtype=torch.bfloat16, device=x_nhwc.device)
    _lib.check(lib.fmc_nhwc_to_cmajor_padded(x_nhwc.data_ptr(), xt.data_ptr(), n, H, W, cin, Lg, G, 3, _stream()),
               "fmc_nhwc_to_cmajor_padded")
    _lib.check(lib.fmc_nhwc_to_cmajor_padded(dy_nhwc.data_ptr(), dyt.data_ptr(), n, H, W, cout, Lk, 0, 1, _stream()),
               "fmc_nhwc_to_cmajor_padded")
    rows = xt.view(3 * cin, Lg)
    tiles = ((3 * cin + 127) // 128) * ((cout + 127) // 128)
    split = 1
    while split < 16 and tiles * split * 2 <= 512 and Lk // 64 >= split * 4:
        split *= 2
    out = []
    for dy in range(3):
        xs = rows[:, G + (dy - 1) * Wp: G + (dy - 1) * Wp + Lk]          # [3 Cin, Lk] view, row stride Lg
        out.append(linear_bf16(xs, dyt, None, None, 1.0, tile=1, split_k=split))      # [dx * Cin + ci][co]
    dwt = torch.stack(out)                                                # [dy][dx * Cin + ci][co]
    return dwt.view(3, 3, cin, cout).permute(3, 2, 0, 1)                  # logical [Cout, Cin, ky, kx]


class _Conv3x3Trainable(torch.autograd.Function):
    """3x3 / stride 1 / pad 1 conv whose FILTER trains (OMC Adapter, camera encoder): forward, backward-data AND the weight
    gradient on the gfx950 kernels -- forward / dX on the implicit-GEMM kernel (the filter is re-laid-out / flipped per step:
    it changes every step and is small next to the activations), dW as pixel-reduction GEMMs (`conv3x3_weight_grad`)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        with torch.no_grad():
            w_cl = weight.detach().to(x.dtype).contiguous(memory_format=torch.channels_last)
            b = None if bias is None else bias.detach().to(x.dtype)
            return conv3x3(x, w_cl, b)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = dw = db = None
        with torch.no_grad():
            if ctx.needs_input_grad[0]:
                w_flip = weight.detach().to(dy.dtype).flip(2, 3).permute(1, 0, 2, 3).contiguous(memory_format=torch.channels_last)
                dx = conv3x3(dy, w_flip, None)
            if ctx.needs_input_grad[1]:
                dw = conv3x3_weight_grad(x.permute(0, 2, 3, 1), dy.permute(0, 2, 3, 1)).to(weight.dtype)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = dy.sum(dim=(0, 2, 3), dtype=torch.float32).to(weight.dtype)
        return dx, dw, (db if ctx.has_bias else None)


def conv3x3_trainable(x, weight, bias=None):
    return _Conv3x3Trainable.apply(x, weight, bias)
