"""CLIP text encoder (`transformers.CLIPTextModel`, ViT-L/14 text tower of SD-1.5) on the gfx950 stack -- the step BEFORE the denoising
loop (`fmc/pipelines/pipeline_animation_cm_om.py:480-568`: `self.text_encoder(text_input_ids, attention_mask=...)[0]`).

State-dict keys are the transformers ones (`[text_model.]embeddings.{token,position}_embedding.weight`,
`encoder.layers.{i}.{layer_norm1,self_attn.{q,k,v,out}_proj,layer_norm2,mlp.{fc1,fc2}}`, `final_layer_norm`); `load_state_dict` accepts both
the prefixed (transformers 4.x checkpoints) and the bare (5.x) form.  LayerNorm = `fmc_layernorm_fwd`, the q | k | v projection is ONE fused
`fmc_linear_bf16`, out-projection and fc2 carry their residual in the GEMM epilogue; the 77-token causal attention (12 heads x d = 64) goes through
`scaled_dot_product_attention` (the hand-written attention kernels are built for the U-Net's head sizes and are not causal), quick-GELU is
a torch op.  Runs once per prompt; outside the metric."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from .. import hip_ops as K
from .layers import LayerNorm, Linear, linear_op


class CLIPTextConfig:
    def __init__(self, vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                 max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5, eos_token_id=2, **_unused):
        self.vocab_size, self.hidden_size, self.intermediate_size = vocab_size, hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.max_position_embeddings, self.hidden_act, self.layer_norm_eps = max_position_embeddings, hidden_act, layer_norm_eps
        self.eos_token_id = eos_token_id
        if hidden_act not in ("quick_gelu", "gelu"):
            raise NotImplementedError(hidden_act)


class _Embeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.token_embedding = nn.Embedding(c.vocab_size, c.hidden_size)
        self.position_embedding = nn.Embedding(c.max_position_embeddings, c.hidden_size)

    def forward(self, ids):
        return self.token_embedding(ids) + self.position_embedding.weight[: ids.shape[1]]


class _SelfAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        d = c.hidden_size
        self.heads = c.num_attention_heads
        self.q_proj, self.k_proj, self.v_proj, self.out_proj = Linear(d, d), Linear(d, d), Linear(d, d), Linear(d, d)

    def _fused(self):
        key = tuple((p.data_ptr(), p._version) for p in (self.q_proj.weight, self.k_proj.weight, self.v_proj.weight))
        hit = self.__dict__.get("_qkv")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, torch.cat([self.q_proj.weight, self.k_proj.weight, self.v_proj.weight]).contiguous(),
                       torch.cat([self.q_proj.bias, self.k_proj.bias, self.v_proj.bias]).contiguous())
            self.__dict__["_qkv"] = hit
        return hit[1], hit[2]

    def forward(self, x, residual, mask: Optional[torch.Tensor]):
        b, s, d = x.shape
        w, bias = self._fused()
        qkv = linear_op(x, w, bias)                                                  # [b, s, 3 d]: q | k | v
        if qkv.is_cuda and K.attention_ok(d // self.heads):
            # causal (+ key-padding, `use_attention_mask` configurations) softmax on `fmc_attention_fwd`, q / k / v read in place from the fused projection
            o = K.attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], self.heads, causal=True, key_keep=None if mask is None else mask.bool())
        else:
            q, k, v = (qkv.view(b, s, 3, self.heads, d // self.heads)[:, :, i].transpose(1, 2) for i in range(3))      # [b, heads, s, dh]
            if mask is None:
                o = F.scaled_dot_product_attention(q, k, v, is_causal=True)
            else:                                                                       # causal AND padding mask
                causal = torch.ones(s, s, dtype=torch.bool, device=x.device).tril()
                o = F.scaled_dot_product_attention(q, k, v, attn_mask=causal[None, None] & mask[:, None, None, :].bool())
            o = o.transpose(1, 2).reshape(b, s, d)
        return linear_op(o, self.out_proj.weight, self.out_proj.bias, residual)


class _MLP(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.fc1, self.fc2 = Linear(c.hidden_size, c.intermediate_size), Linear(c.intermediate_size, c.hidden_size)
        self.quick = c.hidden_act == "quick_gelu"

    def forward(self, x, residual):
        h = self.fc1(x)
        h = h * torch.sigmoid(1.702 * h) if self.quick else F.gelu(h)
        return linear_op(h, self.fc2.weight, self.fc2.bias, residual)


class _Layer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layer_norm1 = LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.self_attn = _SelfAttention(c)
        self.layer_norm2 = LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.mlp = _MLP(c)

    def forward(self, x, mask):
        x = self.self_attn(self.layer_norm1(x), x, mask)
        return self.mlp(self.layer_norm2(x), x)


class _Encoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(c) for _ in range(c.num_hidden_layers)])


class CLIPTextOutput(tuple):
    """`out[0]` / `.last_hidden_state`, `out[1]` / `.pooler_output` (the hidden state at each sequence's EOS token)."""

    def __new__(cls, last_hidden_state, pooler_output):
        self = super().__new__(cls, (last_hidden_state, pooler_output))
        self.last_hidden_state, self.pooler_output = last_hidden_state, pooler_output
        return self


class CLIPTextModel(nn.Module):
    def __init__(self, config: CLIPTextConfig):
        super().__init__()
        self.config = config
        self.embeddings = _Embeddings(config)
        self.encoder = _Encoder(config)
        self.final_layer_norm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    @property
    def dtype(self):
        return self.final_layer_norm.weight.dtype

    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, **_unused):
        x = self.embeddings(input_ids).contiguous()
        for layer in self.encoder.layers:
            x = layer(x, attention_mask)
        x = self.final_layer_norm(x)
        if self.config.eos_token_id == 2:       # legacy configs (SD-1.5): the EOS token is the highest id of the sequence
            eos = input_ids.to(torch.int).argmax(dim=-1)
        else:
            eos = (input_ids == self.config.eos_token_id).int().argmax(dim=-1)
        return CLIPTextOutput(x, x[torch.arange(x.shape[0], device=x.device), eos])

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in state_dict.items()
              if not k.endswith("position_ids")}
        return super().load_state_dict(sd, strict=strict, **kw)
