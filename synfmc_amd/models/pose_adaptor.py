"""Camera Encoder and the CMC wrapper (`fmc/models/pose_adaptor.py`) on the gfx950 path.

`CameraPoseEncoder` keeps the reference's constructor, module tree (`encoder_conv_in`,
`encoder_down_conv_blocks.{i}.{j}.{in_conv,block1,block2,down_opt}`,
`encoder_down_attention_blocks.{i}.{j}`...) and `forward(x: b c f h w) -> 4 x (b f) c h w` contract
(pose_adaptor.py:224-240).  Internally everything is channels-last: the 1x1 convs (`ksize=1`, cam.yaml) are
token GEMMs, the temporal blocks attend over the frame axis in place (no `(b f) c h w <-> (b h w) f c`
transposes), and `forward_unshuffled` accepts the Pluecker embedding exactly as `fmc_plucker_fwd(layout=2)`
writes it (PixelUnshuffle(8) already applied).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from .layers import Conv2d, from_tokens, to_tokens
from .motion_module import TemporalTransformerBlock
from .resnet import _frames_first


def get_parameter_dtype(parameter: torch.nn.Module):
    params = tuple(parameter.parameters())
    if len(params) > 0:
        return params[0].dtype
    buffers = tuple(parameter.buffers())
    return buffers[0].dtype


def conv_nd(dims, *args, **kwargs):
    if dims == 2:
        return Conv2d(*args, **kwargs)
    raise ValueError(f"unsupported dimensions: {dims}")


def avg_pool_nd(dims, *args, **kwargs):
    if dims == 2:
        return nn.AvgPool2d(*args, **kwargs)
    raise ValueError(f"unsupported dimensions: {dims}")


class Downsample(nn.Module):
    """stride-2 conv or 2x2 average pool (pose_adaptor.py:75-99)."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        self.use_conv, self.dims = use_conv, dims
        if use_conv:
            self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=2, padding=padding)
        else:
            assert self.channels == self.out_channels
            self.op = avg_pool_nd(dims, kernel_size=2, stride=2)

    def forward(self, x):
        assert x.shape[1] == self.channels
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        return self.op(x)


class ResnetBlock(nn.Module):
    """pose_adaptor.py:102-135 (skip conv, when present, maps `in_c`)."""

    def __init__(self, in_c, out_c, down, ksize=3, sk=False, use_conv=True):
        super().__init__()
        in_c, out_c = int(in_c), int(out_c)
        ps = ksize // 2
        self.in_conv = Conv2d(in_c, out_c, ksize, 1, ps) if (in_c != out_c or sk is False) else None
        self.block1 = Conv2d(out_c, out_c, 3, 1, 1)
        self.act = nn.ReLU()
        self.block2 = Conv2d(out_c, out_c, ksize, 1, ps)
        self.skep = Conv2d(in_c, out_c, ksize, 1, ps) if sk is False else None
        self.down = down
        if self.down:
            self.down_opt = Downsample(in_c, use_conv=use_conv)

    def forward(self, x):
        if self.down:
            x = self.down_opt(x)
        if self.in_conv is not None:
            x = self.in_conv(x)
        h = self.block2(F.relu(self.block1(x)))
        return h + (self.skep(x) if self.skep is not None else x)


class CameraPoseEncoder(nn.Module):
    def __init__(self, downscale_factor, channels=[320, 640, 1280, 1280], nums_rb=3, cin=64, ksize=3, sk=False,
                 use_conv=True, compression_factor=1, temporal_attention_nhead=8,
                 attention_block_types=("Temporal_Self",), temporal_position_encoding=False,
                 temporal_position_encoding_max_len=16, rescale_output_factor=1.0):
        super().__init__()
        self.downscale_factor = downscale_factor
        self.unshuffle = nn.PixelUnshuffle(downscale_factor)
        self.channels, self.nums_rb = channels, nums_rb
        self.encoder_down_conv_blocks = nn.ModuleList()
        self.encoder_down_attention_blocks = nn.ModuleList()
        for i in range(len(channels)):
            conv_layers, attn_layers = nn.ModuleList(), nn.ModuleList()
            mid = int(channels[i] / compression_factor)
            for j in range(nums_rb):
                if j == 0 and i != 0:
                    in_dim, out_dim, down = channels[i - 1], mid, True
                elif j == 0:
                    in_dim, out_dim, down = channels[0], mid, False
                elif j == nums_rb - 1:
                    in_dim, out_dim, down = mid, channels[i], False
                else:
                    in_dim, out_dim, down = mid, mid, False
                conv_layers.append(ResnetBlock(in_dim, out_dim, down=down, ksize=ksize, sk=sk, use_conv=use_conv))
                attn_layers.append(TemporalTransformerBlock(
                    dim=out_dim, num_attention_heads=temporal_attention_nhead,
                    attention_head_dim=int(out_dim / temporal_attention_nhead),
                    attention_block_types=tuple(attention_block_types), dropout=0.0, cross_attention_dim=None,
                    temporal_position_encoding=temporal_position_encoding,
                    temporal_position_encoding_max_len=temporal_position_encoding_max_len,
                    rescale_output_factor=rescale_output_factor))
            self.encoder_down_conv_blocks.append(conv_layers)
            self.encoder_down_attention_blocks.append(attn_layers)
        self.encoder_conv_in = Conv2d(cin, channels[0], 3, 1, 1)

    @property
    def dtype(self) -> torch.dtype:
        return get_parameter_dtype(self)

    def forward(self, x):
        """x: `b c f h w` Pluecker embedding -> list of 4 features `(b f) c h w`."""
        bs = x.shape[0]
        x4, _, _ = _frames_first(x)
        x4 = self.unshuffle(x4)
        return self._encode(x4.to(self.dtype), bs)

    def forward_unshuffled(self, x_cl: torch.Tensor, bs: int):
        """x_cl: `[(b f), h, w, cin]` channels-last, PixelUnshuffle already applied (`fmc_plucker_fwd` layout 2)."""
        return self._encode(x_cl.permute(0, 3, 1, 2).to(self.dtype), bs)

    def _encode(self, x4, bs):
        features = []
        x4 = self.encoder_conv_in(x4)
        for res_block, attention_block in zip(self.encoder_down_conv_blocks, self.encoder_down_attention_blocks):
            for res_layer, attention_layer in zip(res_block, attention_block):
                x4 = res_layer(x4)
                n, c, h, w = x4.shape
                t = to_tokens(x4).view(bs, n // bs, h * w, c)          # native temporal tokens [B, F, (h w), C]
                t = attention_layer(t)
                x4 = from_tokens(t.view(n, h * w, c), h, w)
            features.append(x4)
        return features


def features_to_video(feats, bs):
    """`(b f) c h w` -> `b c f h w` views (no copy on channels-last storage)."""
    out = []
    for x in feats:
        n, c, h, w = x.shape
        t = x.permute(0, 2, 3, 1)
        if not t.is_contiguous():
            t = t.contiguous()
        out.append(t.view(bs, n // bs, h, w, c).permute(0, 4, 1, 2, 3))
    return out


class PoseAdaptor(nn.Module):
    """pose_adaptor.py:56-72."""

    def __init__(self, unet, pose_encoder):
        super().__init__()
        self.unet = unet
        self.pose_encoder = pose_encoder

    def forward(self, noisy_latents, timesteps, encoder_hidden_states, pose_embedding):
        assert pose_embedding.ndim == 5
        bs = pose_embedding.shape[0]
        pose_embedding_features = features_to_video(self.pose_encoder(pose_embedding), bs)
        return self.unet(noisy_latents, timesteps, encoder_hidden_states,
                         pose_embedding_features=pose_embedding_features).sample
