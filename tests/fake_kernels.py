"""CPU stand-ins for `synfmc_amd.hip_ops`, for HOST-LOGIC tests only (`-m "not gpu"`).

The product never uses these: they are monkey-patched over `synfmc_amd.hip_ops` by a pytest fixture so that
the channels-last plumbing, kwargs routing and processor wiring of the product modules can be checked against
the oracle in a container without a GPU.  Each function restates the kernel's contract in plain PyTorch."""
import torch
import torch.nn.functional as F


def groupnorm_silu(x, gamma, beta, groups, eps, act, x2=None, gn_tag=None):
    if x2 is not None:
        x = torch.cat([x, x2], dim=-1)
    y = F.group_norm(x.float().permute(0, 2, 1), groups, gamma.float(), beta.float(), eps)
    y = F.silu(y) if act else y
    return y.permute(0, 2, 1).contiguous().to(x.dtype)


def layernorm(x, gamma, beta, eps=1e-5, pe=None, pe_inner=1, pe_frames=1):
    y = F.layer_norm(x.float(), (x.shape[-1],), gamma.float(), beta.float(), eps)
    if pe is not None:
        rows = torch.arange(x.numel() // x.shape[-1])
        y = y + pe[(rows // pe_inner) % pe_frames].view(*x.shape)
    return y.to(x.dtype)


def geglu(x):
    a, g = x.float().chunk(2, dim=-1)
    return (a * F.gelu(g)).to(x.dtype)


def _attn(q, k, v, heads, scale):
    B, S, C = q.shape
    D = C // heads
    scale = D ** -0.5 if scale is None else scale
    qh = q.float().reshape(B, S, heads, D).permute(0, 2, 1, 3)
    kh = k.float().reshape(B, k.shape[1], heads, D).permute(0, 2, 1, 3)
    vh = v.float().reshape(B, v.shape[1], heads, D).permute(0, 2, 1, 3)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1)
    return (p @ vh).permute(0, 2, 1, 3).reshape(B, S, C)


def spatial_attention(q, k, v, heads, scale=None, return_lse=False):
    rep = q.shape[0] // k.shape[0]
    o = _attn(q, k.repeat_interleave(rep, 0), v.repeat_interleave(rep, 0), heads, scale).to(q.dtype)
    return (o, None) if return_lse else o


def temporal_attention(q, k, v, heads, scale=None):
    if q.ndim == 4:
        B, Fr, P, C = q.shape
        f = lambda t: t.permute(0, 2, 1, 3).reshape(B * P, Fr, C)
        o = _attn(f(q), f(k), f(v), heads, scale)
        return o.reshape(B, P, Fr, C).permute(0, 2, 1, 3).contiguous().to(q.dtype)
    return _attn(q, k, v, heads, scale).to(q.dtype)


def self_attention_qkv(qkv, heads, scale, temporal):
    c = qkv.shape[-1] // 3
    q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
    return temporal_attention(q, k, v, heads, scale) if temporal else spatial_attention(q, k, v, heads, scale)


def cross_attention_q_kv(q, kv, heads, scale):
    c = q.shape[-1]
    return spatial_attention(q, kv[..., :c], kv[..., c:], heads, scale)


def mask_modulate(x, mask_in, h, w):
    m = F.interpolate(mask_in[:, None], size=(h, w), mode="nearest")
    return (x.float() * m.reshape(x.shape[0], h * w, 1)).to(x.dtype), m[:, 0].contiguous()


def feature_add(h, t, inplace=False):
    skip = h.numel() - t.numel()
    if inplace:
        h.reshape(-1)[skip:] += t.reshape(-1)
        return h
    pad = torch.cat([torch.zeros(skip, dtype=t.dtype), t.reshape(-1)]) if skip else t.reshape(-1)
    return h + pad.view_as(h)


def cfg_ddim_step(eps, x, guidance, alpha_t, alpha_prev, has_uncond):
    e = eps.float()
    if has_uncond:
        eu, ec = e.chunk(2)
        e = eu + guidance * (ec - eu)
    x0 = (x - (1 - alpha_t) ** 0.5 * e) / alpha_t ** 0.5
    return alpha_prev ** 0.5 * x0 + (1 - alpha_prev) ** 0.5 * e


def plucker(K, c2w, H, W, layout="bfhwc", dtype=torch.float32):
    from oracle.conditioning import ray_condition
    B, Fr = K.shape[:2]
    if c2w.shape[2] == 3:
        bottom = torch.tensor([0, 0, 0, 1.0]).view(1, 1, 1, 4).expand(B, Fr, 1, 4)
        c2w = torch.cat([c2w, bottom], dim=2)
    out = ray_condition(K.float(), c2w.float(), H, W)
    if layout == "bfhwc":
        return out.to(dtype)
    if layout == "bcfhw":
        return out.permute(0, 4, 1, 2, 3).contiguous().to(dtype)
    return F.pixel_unshuffle(out.permute(0, 1, 4, 2, 3).reshape(B * Fr, 6, H, W), 8).permute(0, 2, 3, 1).contiguous().to(dtype)


def omc_rasterize(poses, masks, layout="planar", dtype=torch.float32):
    BF, n, H, W = masks.shape
    feat = torch.zeros(BF, 13, H, W)
    mo = torch.zeros(BF, 1, H, W)
    for o in range(n):
        m = masks[:, o]
        on = m > 0
        val = torch.cat([poses[:, o].view(BF, 12, 1, 1) * m[:, None], m[:, None]], dim=1)
        feat = torch.where(on[:, None], val, feat)
        mo = torch.where(on[:, None], m[:, None], mo)
    feat = feat * mo
    if layout == "planar":
        return feat.to(dtype), mo
    return F.pixel_unshuffle(feat, 8).permute(0, 2, 3, 1).contiguous().to(dtype), mo[:, 0].contiguous()


ALL = ["groupnorm_silu", "layernorm", "geglu", "spatial_attention", "temporal_attention", "self_attention_qkv",
       "cross_attention_q_kv", "mask_modulate",
       "feature_add", "cfg_ddim_step", "plucker", "omc_rasterize"]


def install(monkeypatch):
    import synfmc_amd.hip_ops as K
    g = globals()
    for name in ALL:
        monkeypatch.setattr(K, name, g[name])
