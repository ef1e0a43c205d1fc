"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): plain-PyTorch fp32 restatement of the DECODER of diffusers 0.24.0's
`AutoencoderKL` as SD-1.5 configures it (`vae/config.json`: block_out_channels 128/256/512/512, layers_per_block 2, norm_num_groups 32,
latent_channels 4; `fmc/pipelines/pipeline_animation_cm_om.py:465-478` calls `self.vae.decode(latents).sample`).
PARITY UNPINNED against diffusers' outputs (diffusers is not installed in the build container and the reference holds no vector for it) -- cross-checked
instead (tests/test_cpu_misc.py::test_vae_restatement_*): every primitive against torch built-ins and against the U-Net restatement's classes on the same
weights, the tree against the published SD-1.5 VAE (83,653,863 parameters, checkpoint key names / shapes).  The module layout and
arithmetic below restate the published implementation (models/vae.py `Decoder`, models/unet_2d_blocks.py `UNetMidBlock2D` /
`UpDecoderBlock2D`, models/resnet.py `ResnetBlock2D(temb_channels=None)` / `Upsample2D`, models/attention_processor.py `Attention` with
`residual_connection=True`, one head, `norm_num_groups=32`).  The CLIP text encoder needs no restatement: `transformers` is installed and
`transformers.CLIPTextModel` itself is the oracle of `synfmc_amd.models.clip_text`."""
import torch
import torch.nn.functional as F
from torch import nn


class Resnet(nn.Module):
    def __init__(self, cin, cout, groups=32, eps=1e-6):
        super().__init__()
        self.norm1, self.conv1 = nn.GroupNorm(groups, cin, eps=eps), nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2, self.conv2 = nn.GroupNorm(groups, cout, eps=eps), nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class Attn(nn.Module):
    def __init__(self, c, groups=32, eps=1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=eps)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def forward(self, x):
        n, c, h, w = x.shape
        t = self.group_norm(x).view(n, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        p = torch.softmax(torch.baddbmm(torch.empty(n, h * w, h * w), q, k.transpose(1, 2), beta=0, alpha=c ** -0.5), dim=-1)
        o = self.to_out[0](torch.bmm(p, v)).transpose(1, 2).reshape(n, c, h, w)
        return o + x


class Up(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Mid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.attentions = nn.ModuleList([Attn(c, groups)])
        self.resnets = nn.ModuleList([Resnet(c, c, groups), Resnet(c, c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class UpBlock(nn.Module):
    def __init__(self, cin, cout, layers, up, groups):
        super().__init__()
        self.resnets = nn.ModuleList([Resnet(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.upsamplers = nn.ModuleList([Up(cout)]) if up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return x if self.upsamplers is None else self.upsamplers[0](x)


class Decoder(nn.Module):
    def __init__(self, cin=4, cout=3, widths=(128, 256, 512, 512), layers_per_block=2, groups=32):
        super().__init__()
        rev = list(reversed(widths))
        self.conv_in = nn.Conv2d(cin, rev[0], 3, padding=1)
        self.mid_block = Mid(rev[0], groups)
        blocks, prev = [], rev[0]
        for i, c in enumerate(rev):
            blocks.append(UpBlock(prev, c, layers_per_block + 1, i != len(rev) - 1, groups))
            prev = c
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(groups, rev[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(rev[-1], cout, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKLDecoderOnly(nn.Module):
    def __init__(self, widths=(128, 256, 512, 512), layers_per_block=2, groups=32, latent_channels=4):
        super().__init__()
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.decoder = Decoder(latent_channels, 3, widths, layers_per_block, groups)

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))


# ---- encoder half (round 4): diffusers models/vae.py `Encoder(double_z=True)`, models/unet_2d_blocks.py `DownEncoderBlock2D`, models/resnet.py
# `Downsample2D(use_conv=True, padding=0)` (pads (0, 1, 0, 1), then 3x3 stride 2), `DiagonalGaussianDistribution`; reached from
# `train_cam_obj_ctrl.py:786` / `train_cam_ctrl.py:544` (`vae.encode(pixel_values).latent_dist.sample() * 0.18215`).  PARITY UNPINNED like the decoder.
class Down(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, layers, down, groups):
        super().__init__()
        self.resnets = nn.ModuleList([Resnet(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Down(cout)]) if down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return x if self.downsamplers is None else self.downsamplers[0](x)


class Encoder(nn.Module):
    def __init__(self, cin=3, latent=4, widths=(128, 256, 512, 512), layers_per_block=2, groups=32):
        super().__init__()
        self.conv_in = nn.Conv2d(cin, widths[0], 3, padding=1)
        blocks, prev = [], widths[0]
        for i, c in enumerate(widths):
            blocks.append(DownBlock(prev, c, layers_per_block, i != len(widths) - 1, groups))
            prev = c
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = Mid(prev, groups)
        self.conv_norm_out = nn.GroupNorm(groups, prev, eps=1e-6)
        self.conv_out = nn.Conv2d(prev, 2 * latent, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(self.mid_block(x))))


class AutoencoderKLFull(AutoencoderKLDecoderOnly):
    def __init__(self, widths=(128, 256, 512, 512), layers_per_block=2, groups=32, latent_channels=4):
        super().__init__(widths, layers_per_block, groups, latent_channels)
        self.encoder = Encoder(3, latent_channels, widths, layers_per_block, groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)

    def encode_moments(self, x):
        """(mean, logvar clamped to [-30, 20]) of the posterior."""
        mean, logvar = self.quant_conv(self.encoder(x)).chunk(2, dim=1)
        return mean, logvar.clamp(-30.0, 20.0)
