"""Inflated 2-D layers (`fmc/models/resnet.py:16-37`).  Only `InflatedConv3d` and `InflatedGroupNorm` are on
the hot path; `FusionBlock2D` / `ResnetBlock3D` / `Upsample3D` / `Downsample3D` / `Mish` of the reference are
dead code and are not built."""
from __future__ import annotations

import torch
from torch import nn

from .layers import Conv2d, GroupNorm


def _frames_first(x: torch.Tensor):
    """`b c f h w` -> `(b f) c h w` view over channels-last storage (a copy only if the input is not
    `channels_last_3d`)."""
    b, c, f, h, w = x.shape
    t = x.permute(0, 2, 3, 4, 1)
    if not t.is_contiguous():
        t = t.contiguous()
    y = t.view(b * f, h, w, c).permute(0, 3, 1, 2)
    tag = getattr(x, "_fmc_gn", None)
    if tag is not None:
        y._fmc_gn = tag
    return y, b, f


def _frames_back(y: torch.Tensor, b: int, f: int):
    n, c, h, w = y.shape
    t = y.permute(0, 2, 3, 1)
    if not t.is_contiguous():
        t = t.contiguous()
    out = t.view(b, f, h, w, c).permute(0, 4, 1, 2, 3)
    tag = getattr(y, "_fmc_gn", None)
    if tag is not None:
        out._fmc_gn = tag
    return out


class InflatedConv3d(Conv2d):
    """Per-frame 2-D convolution on a `b c f h w` video (resnet.py:16-24)."""

    def forward(self, x):
        x4, b, f = _frames_first(x)
        return _frames_back(super().forward(x4), b, f)


class InflatedGroupNorm(GroupNorm):
    """Per-frame GroupNorm on a `b c f h w` video (resnet.py:27-37)."""

    def forward(self, x):
        x4, b, f = _frames_first(x)
        return _frames_back(super().forward(x4), b, f)
