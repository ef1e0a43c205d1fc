"""diffusers' `AutoencoderKL` (SD-1.5 VAE) on the gfx950 stack: the decoder -- the step AFTER the denoising loop
(`fmc/pipelines/pipeline_animation_cm_om.py:465-478`: `self.vae.decode(latents[i:i+1]).sample`, frame by frame) -- and, round 4, the encoder --
the step BEFORE the training step (`train_cam_obj_ctrl.py:786`, `train_cam_ctrl.py:544`: `vae.encode(pixel_values).latent_dist.sample() * 0.18215`);
SURVEY.md section 8 f4.

Same sub-module / parameter names as the diffusers class (`post_quant_conv`, `decoder.conv_in`, `decoder.mid_block.{resnets,attentions}`,
`decoder.up_blocks.{i}.{resnets,upsamplers}`, `decoder.conv_norm_out`, `decoder.conv_out`), so an SD-1.5 `vae/diffusion_pytorch_model.*`
state dict loads as it is (`encoder.*`, `quant_conv.*` since round 4; `load_decoder_state_dict` still takes the decoder half alone).

What runs where: GroupNorm(+SiLU) = `fmc_groupnorm_silu_fwd`; every 3x3 convolution = `fmc_conv3x3_bf16` with the residual in its epilogue
and the nearest-2x upsample folded into its operand addressing (the two edge convolutions, `conv_in` 4 -> 512 and `conv_out` 128 -> 3, with
zero-padded channels); the mid-block attention's projections = `fmc_linear_bf16`; its single-head d = 512 softmax(QK^T)V goes through
`torch.nn.functional.scaled_dot_product_attention` (the hand-written attention kernels are built for the U-Net's d = 40 / 80 / 160).
Outside the metric (BASELINE.json: VAE / CLIP excluded)."""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn.functional as F
from torch import nn

from .. import hip_ops as K
from .layers import Conv2d, GroupNorm, Linear, ResnetBlock2D, Upsample2D, from_tokens, linear_op, to_tokens


def _padded_conv3x3(conv: Conv2d, x: torch.Tensor, act_dtype_ok: bool = True) -> torch.Tensor:
    """The two edge convolutions (conv_in 4 -> C, conv_out C -> 3) on the hand-written implicit-GEMM kernel (`Conv2d.padded_conv3x3`)."""
    if not x.is_cuda or torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad):
        return F.conv2d(x, conv.weight, conv.bias, 1, 1)
    return conv.padded_conv3x3(x)


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


class VaeAttention(nn.Module):
    """diffusers `Attention(512, heads=1, dim_head=512, norm_num_groups=32, residual_connection=True, bias=True)` as the VAE mid block
    builds it: GroupNorm -> q, k, v -> softmax(q k^T / sqrt(d)) v -> to_out.0 -> + input."""

    def __init__(self, channels: int, groups: int = 32, eps: float = 1e-6):
        super().__init__()
        self.group_norm = GroupNorm(num_groups=groups, num_channels=channels, eps=eps, affine=True)
        self.to_q, self.to_k, self.to_v = Linear(channels, channels), Linear(channels, channels), Linear(channels, channels)
        self.to_out = nn.ModuleList([Linear(channels, channels), nn.Dropout(0.0)])
        self.scale = channels ** -0.5

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n, c, h, w = x.shape
        res = to_tokens(x)                                             # [n, h w, c]
        t = to_tokens(self.group_norm(x))
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        if t.is_cuda and K.attention_ok(c):
            o = K.attention(q, k, v, 1, self.scale)                    # single head of width c (512 in SD's VAE): `fmc_attention_fwd`
        else:                                                          # widths outside {32 .. 512}: torch (no such VAE ships with FMC)
            o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None], scale=self.scale)[:, 0].contiguous()
        return from_tokens(linear_op(o, self.to_out[0].weight, self.to_out[0].bias, res), h, w)


class _MidBlock(nn.Module):
    def __init__(self, c, groups, eps):
        super().__init__()
        self.attentions = nn.ModuleList([VaeAttention(c, groups, eps)])
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels=c, out_channels=c, temb_channels=None, groups=groups, eps=eps)
                                      for _ in range(2)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x, None)), None)


class _UpDecoderBlock(nn.Module):
    def __init__(self, cin, cout, layers, add_upsample, groups, eps):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels=cin if i == 0 else cout, out_channels=cout, temb_channels=None,
                                                    groups=groups, eps=eps) for i in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout, use_conv=True, out_channels=cout)]) if add_upsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x, None)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class Decoder(nn.Module):
    def __init__(self, in_channels=4, out_channels=3, block_out_channels: Sequence[int] = (128, 256, 512, 512), layers_per_block=2,
                 norm_num_groups=32, eps=1e-6):
        super().__init__()
        rev = list(reversed(block_out_channels))
        self.conv_in = Conv2d(in_channels, rev[0], kernel_size=3, stride=1, padding=1)
        self.mid_block = _MidBlock(rev[0], norm_num_groups, eps)
        blocks, prev = [], rev[0]
        for i, c in enumerate(rev):
            blocks.append(_UpDecoderBlock(prev, c, layers_per_block + 1, i != len(rev) - 1, norm_num_groups, eps))
            prev = c
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = GroupNorm(num_groups=norm_num_groups, num_channels=rev[-1], eps=eps, affine=True)
        self.conv_act = nn.SiLU()
        self.conv_out = Conv2d(rev[-1], out_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, z):
        x = _padded_conv3x3(self.conv_in, z)
        x = self.mid_block(x)
        for b in self.up_blocks:
            x = b(x)
        return _padded_conv3x3(self.conv_out, self.conv_norm_out(x, act=True))


class _Downsample(nn.Module):
    """diffusers `Downsample2D(c, use_conv=True, padding=0)` as `DownEncoderBlock2D` builds it: zero-pad one row / column at the BOTTOM / RIGHT,
    then a 3x3 stride-2 convolution without padding: out(y, x) = sum_t w[t] in(2 y + dy, 2 x + dx), dy, dx in 0..2.  The implicit-GEMM kernel's
    stride-2 mode pads symmetrically (taps 2 y + dy - 1): fed the input shifted by one pixel (zero-padded on all four sides) its output row
    y + 1 / column x + 1 is exactly out(y, x) -- one padded copy of the input, the first output row and column dropped."""

    def __init__(self, channels):
        super().__init__()
        self.conv = Conv2d(channels, channels, kernel_size=3, stride=2, padding=0)

    def forward(self, x):
        n, c, h, w = x.shape
        if (x.is_cuda and x.dtype == torch.bfloat16 and not torch.is_grad_enabled() and h % 2 == 0 and w % 2 == 0 and c % 64 == 0):
            xp = F.pad(x, (1, 1, 1, 1)).contiguous(memory_format=torch.channels_last)
            y = K.conv3x3(xp, self.conv._weight_cl(), self.conv.bias, None, None, (2, 2), (1, 1), own_only=True)
            return y[:, :, 1:, 1:]
        return F.conv2d(F.pad(x, (0, 1, 0, 1)), self.conv.weight, self.conv.bias, 2, 0)


class _DownEncoderBlock(nn.Module):
    def __init__(self, cin, cout, layers, add_downsample, groups, eps):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels=cin if i == 0 else cout, out_channels=cout, temb_channels=None,
                                                    groups=groups, eps=eps) for i in range(layers)])
        self.downsamplers = nn.ModuleList([_Downsample(cout)]) if add_downsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x, None)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class Encoder(nn.Module):
    """diffusers `Encoder(in_channels=3, out_channels=latent, double_z=True)`: conv_in -> 4 x DownEncoderBlock2D (2 ResNet blocks each, stride-2
    conv between levels) -> mid block (ResNet, single-head attention, ResNet) -> GroupNorm + SiLU -> conv_out to 2 x latent channels."""

    def __init__(self, in_channels=3, out_channels=4, block_out_channels: Sequence[int] = (128, 256, 512, 512), layers_per_block=2,
                 norm_num_groups=32, eps=1e-6):
        super().__init__()
        self.conv_in = Conv2d(in_channels, block_out_channels[0], kernel_size=3, stride=1, padding=1)
        blocks, prev = [], block_out_channels[0]
        for i, c in enumerate(block_out_channels):
            blocks.append(_DownEncoderBlock(prev, c, layers_per_block, i != len(block_out_channels) - 1, norm_num_groups, eps))
            prev = c
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = _MidBlock(prev, norm_num_groups, eps)
        self.conv_norm_out = GroupNorm(num_groups=norm_num_groups, num_channels=prev, eps=eps, affine=True)
        self.conv_act = nn.SiLU()
        self.conv_out = Conv2d(prev, 2 * out_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        x = _padded_conv3x3(self.conv_in, x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(self.conv_norm_out(x, act=True))


class DiagonalGaussianDistribution:
    """diffusers `DiagonalGaussianDistribution(moments)`: mean | logvar along the channel axis, logvar clamped to [-30, 20]."""

    def __init__(self, parameters: torch.Tensor, deterministic: bool = False):
        self.parameters = parameters
        self.deterministic = deterministic
        self.mean, logvar = torch.chunk(parameters.float(), 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """The random stream is diffusers' `randn_tensor(shape, generator, device=parameters.device, dtype=parameters.dtype)`: drawn in the
        PARAMETERS' dtype (a bf16 model draws bf16 noise), on the generator's device when that is the CPU and the parameters live on the GPU."""
        p = self.parameters
        gen_dev = p.device if generator is None else generator.device
        draw_dev = gen_dev if (gen_dev.type == "cpu" and p.device.type != "cpu") else p.device
        noise = torch.randn(self.mean.shape, generator=generator, device=draw_dev, dtype=p.dtype).to(p.device)
        return (self.mean + self.std * noise.float()).to(p.dtype)

    def kl(self, other: "Optional[DiagonalGaussianDistribution]" = None) -> torch.Tensor:
        if self.deterministic:
            return torch.zeros(1, device=self.parameters.device)
        if other is None:
            return 0.5 * torch.sum(self.mean.pow(2) + self.var - 1.0 - self.logvar, dim=[1, 2, 3])
        return 0.5 * torch.sum((self.mean - other.mean).pow(2) / other.var + self.var / other.var - 1.0 - self.logvar + other.logvar,
                               dim=[1, 2, 3])

    def nll(self, sample: torch.Tensor, dims=(1, 2, 3)) -> torch.Tensor:
        if self.deterministic:
            return torch.zeros(1, device=self.parameters.device)
        log2pi = 1.8378770664093453
        return 0.5 * torch.sum(log2pi + self.logvar + (sample.float() - self.mean).pow(2) / self.var, dim=list(dims))

    def mode(self) -> torch.Tensor:
        return self.mean.to(self.parameters.dtype)


class AutoencoderKLOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class AutoencoderKL(nn.Module):
    """`AutoencoderKL(...).decode(z).sample` (`[N, 4, h, w]` latents already divided by the scaling factor -> `[N, 3, 8h, 8w]`)."""

    def __init__(self, in_channels=3, out_channels=3, block_out_channels: Sequence[int] = (128, 256, 512, 512), layers_per_block=2,
                 latent_channels=4, norm_num_groups=32, scaling_factor=0.18215, **_unused):
        super().__init__()
        self.scaling_factor = scaling_factor
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.quant_conv = Conv2d(2 * latent_channels, 2 * latent_channels, kernel_size=1)
        self.post_quant_conv = Conv2d(latent_channels, latent_channels, kernel_size=1)
        self.decoder = Decoder(latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups)

    @property
    def dtype(self):
        return self.post_quant_conv.weight.dtype

    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """`vae.encode(pixel_values).latent_dist.sample()` (train_cam_obj_ctrl.py:786): `[N, 3, H, W]` in [-1, 1] -> the posterior over
        `[N, 4, H / 8, W / 8]` latents (the caller multiplies the sample by `scaling_factor`)."""
        x = x.to(self.dtype)
        if x.is_cuda and not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        moments = F.conv2d(self.encoder(x), self.quant_conv.weight, self.quant_conv.bias)
        dist = DiagonalGaussianDistribution(moments)
        return AutoencoderKLOutput(dist) if return_dict else (dist,)

    def decode(self, z: torch.Tensor, return_dict: bool = True):
        z = z.to(self.dtype)
        x = self.decoder(F.conv2d(z, self.post_quant_conv.weight, self.post_quant_conv.bias))
        return DecoderOutput(x) if return_dict else (x,)

    def load_decoder_state_dict(self, state_dict, strict: bool = True):
        """Load a full AutoencoderKL state dict, ignoring the encoder / quant_conv halves FMC never runs."""
        sd = {k: v for k, v in state_dict.items() if k.startswith(("decoder.", "post_quant_conv."))}
        missing, unexpected = self.load_state_dict(sd, strict=False)
        missing = [k for k in missing if not k.startswith(("encoder.", "quant_conv."))]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_decoder_state_dict: missing {missing}, unexpected {unexpected}")
        return missing, unexpected
