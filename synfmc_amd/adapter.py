"""Object Encoder (`fmc/adapter.py:109-192`, the OMC `Adapter`) on the gfx950 path.

Same constructor, module tree (`conv_in`, `zero_conv_in`, `body.{i}.{in_conv,block1,block2,down_opt}`,
`zero_conv_out_list.{i}`) and `forward(x, mask_feat) -> 4 x (b f) c h w`.  Channels-last inside; the per-level
`F.interpolate(mask, nearest)` + `mask * x` (reference :175-177, a cascaded mask pyramid) is one
`fmc_mask_modulate_fwd` pass per level that also emits the next level's mask.  `StyleAdapter`,
`Adapter_light` and `extractor` of the reference file are unused by FMC and are not built.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from . import hip_ops as K
from .models.layers import Conv2d, from_tokens, to_tokens


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        assert dims == 2
        self.channels, self.out_channels, self.use_conv = channels, out_channels or channels, use_conv
        if use_conv:
            self.op = Conv2d(self.channels, self.out_channels, 3, stride=2, padding=padding)
        else:
            assert self.channels == self.out_channels
            self.op = nn.AvgPool2d(kernel_size=2, stride=2)

    def forward(self, x):
        assert x.shape[1] == self.channels
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        return self.op(x)


class ResnetBlock(nn.Module):
    """adapter.py:64-98 (skip conv, when present, maps `out_c`)."""

    def __init__(self, in_c, out_c, down, ksize=3, sk=False, use_conv=True):
        super().__init__()
        ps = ksize // 2
        self.in_conv = Conv2d(in_c, out_c, ksize, 1, ps) if (in_c != out_c or sk is False) else None
        self.block1 = Conv2d(out_c, out_c, 3, 1, 1)
        self.act = nn.ReLU()
        self.block2 = Conv2d(out_c, out_c, ksize, 1, ps)
        self.skep = Conv2d(out_c, out_c, ksize, 1, ps) if sk is False else None
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


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


class Adapter(nn.Module):
    def __init__(self, channels=[320, 640, 1280, 1280], nums_rb=3, cin=64, ksize=3, sk=False, use_conv=True,
                 align_training_size=0, use_pre_zero_conv=False, use_post_zero_conv=False):
        super().__init__()
        if align_training_size != 0:
            raise NotImplementedError("align_training_size > 0 ends in `assert False` in the reference (adapter.py:182)")
        self.align_training_size = align_training_size
        self.unshuffle = nn.PixelUnshuffle(8)
        self.channels, self.nums_rb = channels, nums_rb
        body = []
        for i in range(len(channels)):
            for j in range(nums_rb):
                if (i != 0) and (j == 0):
                    body.append(ResnetBlock(channels[i - 1], channels[i], down=True, ksize=ksize, sk=sk, use_conv=use_conv))
                else:
                    body.append(ResnetBlock(channels[i], channels[i], down=False, ksize=ksize, sk=sk, use_conv=use_conv))
        self.body = nn.ModuleList(body)
        self.conv_in = Conv2d(cin, channels[0], 3, 1, 1)
        self.zero_conv_in = zero_module(Conv2d(cin, cin, kernel_size=1, stride=1, padding=0)) \
            if use_pre_zero_conv else nn.Identity()
        self.zero_conv_out_list = nn.ModuleList([
            zero_module(Conv2d(c, c, kernel_size=1, stride=1, padding=0)) if use_post_zero_conv else nn.Identity()
            for c in channels])

    def forward(self, x, mask_feat):
        """x `[(b f), 13, H, W]`, mask_feat `[(b f), 1, H, W]` or None (the reference contract) -- or, when
        `mask_feat` is 3-D `[(b f), H, W]`, x is `[(b f), H/8, W/8, cin]` as `fmc_omc_rasterize_fwd(layout=2)` writes it."""
        if mask_feat is not None and mask_feat.ndim == 3:
            return self.forward_unshuffled(x, mask_feat)
        dtype = self.conv_in.weight.dtype
        return self._encode(self.unshuffle(x).to(dtype), None if mask_feat is None else mask_feat[:, 0])

    def forward_unshuffled(self, x_cl, mask):
        """x_cl `[(b f), H/8, W/8, cin]` channels-last with PixelUnshuffle(8) applied and mask `[(b f), H, W]` fp32:
        exactly what `fmc_omc_rasterize_fwd(layout=2)` writes."""
        return self._encode(x_cl.permute(0, 3, 1, 2).to(self.conv_in.weight.dtype), mask)

    def _encode(self, x, mask):
        features = []
        x = self.conv_in(self.zero_conv_in(x))
        if mask is not None:
            mask = mask.to(torch.float32).contiguous()
        for i in range(len(self.channels)):
            for j in range(self.nums_rb):
                x = self.body[i * self.nums_rb + j](x)
            x = self.zero_conv_out_list[i](x)
            if mask is not None:
                n, c, h, w = x.shape
                y, mask = K.mask_modulate(to_tokens(x), mask, h, w)
                x = from_tokens(y, h, w)
            features.append(x)
        return features
