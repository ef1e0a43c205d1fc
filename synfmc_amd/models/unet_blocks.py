"""3-D U-Net blocks of FMC (`fmc/models/unet_blocks.py`): same five classes, factories, constructor
arguments, sub-module names and forward signatures.  A layer is resnet -> [spatial transformer] ->
[motion module]; tensors are logically `b c f h w` and physically channels-last, so the nine
`einops.rearrange` transposing copies per layer of the reference are views here.

The OMC injection (`hidden += traj_features[self.traj_fea_idx]`, `fmc/modified_modules.py:115-117`) lives in
`_DownBlock._down`; the stock `forward`s never inject (as in the reference, where `traj_features` reaching
the un-patched block is an error) -- `synfmc_amd.modified_modules.Adapted_*_forward` enable it.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .. import hip_ops as K
from .layers import Downsample2D, ResnetBlock2D, Transformer2DModel, Upsample2D
from .motion_module import get_motion_module
from .resnet import _frames_back, _frames_first


def get_down_block(down_block_type, num_layers, in_channels, out_channels, temb_channels, add_downsample, resnet_eps,
                   resnet_act_fn, attn_num_head_channels, resnet_groups=None, cross_attention_dim=None,
                   downsample_padding=None, dual_cross_attention=False, use_linear_projection=False,
                   only_cross_attention=False, upcast_attention=False, resnet_time_scale_shift="default",
                   use_motion_module=None, motion_module_type=None, motion_module_kwargs=None):
    down_block_type = down_block_type[7:] if down_block_type.startswith("UNetRes") else down_block_type
    common = dict(num_layers=num_layers, in_channels=in_channels, out_channels=out_channels,
                  temb_channels=temb_channels, add_downsample=add_downsample, resnet_eps=resnet_eps,
                  resnet_act_fn=resnet_act_fn, resnet_groups=resnet_groups, downsample_padding=downsample_padding,
                  resnet_time_scale_shift=resnet_time_scale_shift, use_motion_module=use_motion_module,
                  motion_module_type=motion_module_type, motion_module_kwargs=motion_module_kwargs)
    if down_block_type == "DownBlock3D":
        return DownBlock3D(**common)
    if down_block_type == "CrossAttnDownBlock3D":
        if cross_attention_dim is None:
            raise ValueError("cross_attention_dim must be specified for CrossAttnDownBlock3D")
        return CrossAttnDownBlock3D(cross_attention_dim=cross_attention_dim,
                                    attn_num_head_channels=attn_num_head_channels,
                                    dual_cross_attention=dual_cross_attention,
                                    use_linear_projection=use_linear_projection,
                                    only_cross_attention=only_cross_attention, upcast_attention=upcast_attention,
                                    **common)
    raise ValueError(f"{down_block_type} does not exist.")


def get_up_block(up_block_type, num_layers, in_channels, out_channels, prev_output_channel, temb_channels,
                 add_upsample, resnet_eps, resnet_act_fn, attn_num_head_channels, resnet_groups=None,
                 cross_attention_dim=None, dual_cross_attention=False, use_linear_projection=False,
                 only_cross_attention=False, upcast_attention=False, resnet_time_scale_shift="default",
                 use_motion_module=None, motion_module_type=None, motion_module_kwargs=None):
    up_block_type = up_block_type[7:] if up_block_type.startswith("UNetRes") else up_block_type
    common = dict(num_layers=num_layers, in_channels=in_channels, out_channels=out_channels,
                  prev_output_channel=prev_output_channel, temb_channels=temb_channels, add_upsample=add_upsample,
                  resnet_eps=resnet_eps, resnet_act_fn=resnet_act_fn, resnet_groups=resnet_groups,
                  resnet_time_scale_shift=resnet_time_scale_shift, use_motion_module=use_motion_module,
                  motion_module_type=motion_module_type, motion_module_kwargs=motion_module_kwargs)
    if up_block_type == "UpBlock3D":
        return UpBlock3D(**common)
    if up_block_type == "CrossAttnUpBlock3D":
        if cross_attention_dim is None:
            raise ValueError("cross_attention_dim must be specified for CrossAttnUpBlock3D")
        return CrossAttnUpBlock3D(cross_attention_dim=cross_attention_dim,
                                  attn_num_head_channels=attn_num_head_channels,
                                  dual_cross_attention=dual_cross_attention,
                                  use_linear_projection=use_linear_projection,
                                  only_cross_attention=only_cross_attention, upcast_attention=upcast_attention,
                                  **common)
    raise ValueError(f"{up_block_type} does not exist.")


# ----------------------------------------------------------------------------
# shared machinery
# ----------------------------------------------------------------------------
def _mk_resnet(cin, cout, temb, eps, groups, act, dropout, tss, scale=1.0, pre_norm=True):
    return ResnetBlock2D(in_channels=cin, out_channels=cout, temb_channels=temb, eps=eps, groups=groups,
                         dropout=dropout, time_embedding_norm=tss, non_linearity=act, output_scale_factor=scale,
                         pre_norm=pre_norm)


def _mk_transformer(heads, channels, cross_dim, groups, use_linear_projection, only_cross, upcast):
    return Transformer2DModel(heads, channels // heads, in_channels=channels, num_layers=1,
                              cross_attention_dim=cross_dim, norm_num_groups=groups,
                              use_linear_projection=use_linear_projection, only_cross_attention=only_cross,
                              upcast_attention=bool(upcast))


def _mk_motion(channels, use, mtype, mkw):
    return get_motion_module(in_channels=channels, motion_module_type=mtype, motion_module_kwargs=mkw) if use else None


class _Block3D(nn.Module):
    has_cross_attention = False

    def _check_ckpt(self):
        if self.training and getattr(self, "gradient_checkpointing", False):
            raise NotImplementedError            # unet_blocks.py:378-379, 507-508, 662-663, 788-789

    def _motion_kwargs(self, motion_cross_attention_kwargs):
        kw = motion_cross_attention_kwargs
        ms = getattr(self, "motion_lora_scale", None)
        if ms is not None:
            kw = {"scale": ms} if kw is None else {**kw, "scale": ms}
        return {} if kw is None else kw

    def _layer(self, i, x4, b, f, temb_rep, text, cross_kw, motion_kw, skip4=None, cfg_expand=False):
        """x4: `(b f) c h w`.  resnet -> [transformer] -> [motion module].  `skip4`: up-block skip connection, the
        resnet input is `cat([x4, skip4], 1)`.  `cfg_expand`: x4 is one copy of a CFG batch's identical halves (b = the FULL batch)."""
        if cfg_expand and temb_rep is not None:
            temb_rep = temb_rep[: x4.shape[0]]
        x4 = self.resnets[i](x4, temb_rep, skip=skip4)
        if self.has_cross_attention:
            x4 = self.attentions[i](x4, encoder_hidden_states=text, cross_attention_kwargs=cross_kw,
                                    **({"cfg_expand": True} if cfg_expand else {})).sample
        elif cfg_expand:
            x4 = torch.cat([x4, x4], dim=0)
        mm = self.motion_modules[i] if len(self.motion_modules) else None
        if mm is not None:
            x5 = mm(_frames_back(x4, b, f), encoder_hidden_states=text, cross_attention_kwargs=motion_kw)
            x4 = _frames_first(x5)[0]
        return x4


class _DownBlock(_Block3D):
    def _build(self, in_channels, out_channels, temb_channels, dropout, num_layers, resnet_eps,
               resnet_time_scale_shift, resnet_act_fn, resnet_groups, resnet_pre_norm, output_scale_factor,
               add_downsample, downsample_padding, use_motion_module, motion_module_type, motion_module_kwargs,
               attn=None):
        resnets, attentions, motion_modules = [], [], []
        for i in range(num_layers):
            resnets.append(_mk_resnet(in_channels if i == 0 else out_channels, out_channels, temb_channels, resnet_eps,
                                      resnet_groups, resnet_act_fn, dropout, resnet_time_scale_shift,
                                      output_scale_factor, resnet_pre_norm))
            if attn is not None:
                attentions.append(_mk_transformer(attn["heads"], out_channels, attn["cross_dim"], resnet_groups,
                                                  attn["linear"], attn["only_cross"], attn["upcast"]))
            motion_modules.append(_mk_motion(out_channels, use_motion_module, motion_module_type, motion_module_kwargs))
        if attn is not None:
            self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)
        self.motion_modules = nn.ModuleList(motion_modules) if use_motion_module else motion_modules
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, use_conv=True, out_channels=out_channels,
                                                        padding=downsample_padding, name="op")]) \
            if add_downsample else None
        self.gradient_checkpointing = False

    def _down(self, hidden_states, temb, text, cross_kw, motion_kw, traj_features):
        self._check_ckpt()
        x4, b, f = _frames_first(hidden_states)
        temb_rep = None if self.resnets[0]._t_pre is not None else temb.repeat_interleave(f, dim=0)
        # shared CFG prefix (set by the U-Net for its first down block only): the input holds ONE copy of the two identical halves
        cfg_half = bool(self.__dict__.pop("_cfg_half_input", False))
        if cfg_half:
            b = 2 * b
        outs = ()
        for i in range(len(self.resnets)):
            x4 = self._layer(i, x4, b, f, temb_rep, text, cross_kw, motion_kw, cfg_expand=cfg_half and i == 0)
            outs += (_frames_back(x4, b, f),)
        if traj_features is not None:                      # modified_modules.py:115-117 / 172-174
            t = traj_features[self.traj_fea_idx]
            hb = outs[-1].shape[0]
            if tuple(t.shape[1:]) != tuple(outs[-1].shape[1:]) or t.shape[0] not in (hb, hb // 2) or (t.shape[0] != hb and hb % 2):
                # the reference's `hidden_states + traj_features[idx]` raises on such a mismatch; here the features may cover
                # the whole batch or exactly its conditioned (second) half under classifier-free guidance -- nothing else
                raise ValueError(f"traj_features[{self.traj_fea_idx}] of shape {tuple(t.shape)} does not match hidden states "
                                 f"{tuple(outs[-1].shape)} (batch must equal {hb}, or {hb // 2} = the conditioned CFG half)")
            h5 = outs[-1].permute(0, 2, 3, 4, 1)           # [B, F, h, w, C] storage order, contiguous
            t5 = t.permute(0, 2, 3, 4, 1)
            if not t5.is_contiguous():
                t5 = t5.contiguous()
            if t5.dtype != h5.dtype:
                t5 = t5.to(h5.dtype)
            s5 = K.feature_add(h5, t5)
            x5 = s5.permute(0, 4, 1, 2, 3)
            outs = outs[:-1] + (x5,)
            x4 = _frames_first(x5)[0]
        if self.downsamplers is not None:
            for ds in self.downsamplers:
                x4 = ds(x4)
            outs += (_frames_back(x4, b, f),)
        return _frames_back(x4, b, f), outs


class CrossAttnDownBlock3D(_DownBlock):
    """unet_blocks.py:268-426."""

    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, dropout: float = 0.0,
                 num_layers: int = 1, resnet_eps: float = 1e-6, resnet_time_scale_shift: str = "default",
                 resnet_act_fn: str = "swish", resnet_groups: int = 32, resnet_pre_norm: bool = True,
                 attn_num_head_channels=1, cross_attention_dim=1280, output_scale_factor=1.0, downsample_padding=1,
                 add_downsample=True, dual_cross_attention=False, use_linear_projection=False,
                 only_cross_attention=False, upcast_attention=False, use_motion_module=None, motion_module_type=None,
                 motion_module_kwargs=None):
        super().__init__()
        if dual_cross_attention:
            raise NotImplementedError
        self.has_cross_attention = True
        self.attn_num_head_channels = attn_num_head_channels
        self._build(in_channels, out_channels, temb_channels, dropout, num_layers, resnet_eps,
                    resnet_time_scale_shift, resnet_act_fn, resnet_groups, resnet_pre_norm, output_scale_factor,
                    add_downsample, downsample_padding, use_motion_module, motion_module_type, motion_module_kwargs,
                    attn=dict(heads=attn_num_head_channels, cross_dim=cross_attention_dim, linear=use_linear_projection,
                              only_cross=only_cross_attention, upcast=upcast_attention))

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None,
                motion_module_alpha=1., cross_attention_kwargs={}, motion_cross_attention_kwargs={}):
        cross_kw = dict(cross_attention_kwargs or {})
        if "traj_features" in cross_kw:
            raise TypeError("__call__() got an unexpected keyword argument 'traj_features' (the stock block does not "
                            "consume OMC features: patch it with fmc.modified_modules.Adapted_CrossAttnDownBlock3D_forward)")
        ls = getattr(self, "lora_scale", None)
        if ls is not None:
            cross_kw["scale"] = ls
        return self._down(hidden_states, temb, encoder_hidden_states, cross_kw,
                          self._motion_kwargs(motion_cross_attention_kwargs), None)


class DownBlock3D(_DownBlock):
    """unet_blocks.py:429-540."""

    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, dropout: float = 0.0,
                 num_layers: int = 1, resnet_eps: float = 1e-6, resnet_time_scale_shift: str = "default",
                 resnet_act_fn: str = "swish", resnet_groups: int = 32, resnet_pre_norm: bool = True,
                 output_scale_factor=1.0, add_downsample=True, downsample_padding=1, use_motion_module=None,
                 motion_module_type=None, motion_module_kwargs=None):
        super().__init__()
        self._build(in_channels, out_channels, temb_channels, dropout, num_layers, resnet_eps,
                    resnet_time_scale_shift, resnet_act_fn, resnet_groups, resnet_pre_norm, output_scale_factor,
                    add_downsample, downsample_padding, use_motion_module, motion_module_type, motion_module_kwargs)

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, motion_module_alpha=1.,
                motion_cross_attention_kwargs={}, **kwargs):
        return self._down(hidden_states, temb, encoder_hidden_states, None,
                          self._motion_kwargs(motion_cross_attention_kwargs), None)


class UNetMidBlock3DCrossAttn(_Block3D):
    """unet_blocks.py:144-265: resnet0, then per layer (transformer, [motion module], resnet)."""

    def __init__(self, in_channels: int, temb_channels: int, dropout: float = 0.0, num_layers: int = 1,
                 resnet_eps: float = 1e-6, resnet_time_scale_shift: str = "default", resnet_act_fn: str = "swish",
                 resnet_groups: int = 32, resnet_pre_norm: bool = True, attn_num_head_channels=1,
                 output_scale_factor=1.0, cross_attention_dim=1280, dual_cross_attention=False,
                 use_linear_projection=False, upcast_attention=False, use_motion_module=None,
                 motion_module_type=None, motion_module_kwargs=None):
        super().__init__()
        if dual_cross_attention:
            raise NotImplementedError
        self.has_cross_attention = True
        self.attn_num_head_channels = attn_num_head_channels
        resnet_groups = resnet_groups if resnet_groups is not None else min(in_channels // 4, 32)

        def res():
            return _mk_resnet(in_channels, in_channels, temb_channels, resnet_eps, resnet_groups, resnet_act_fn,
                              dropout, resnet_time_scale_shift, output_scale_factor, resnet_pre_norm)
        resnets, attentions, motion_modules = [res()], [], []
        for _ in range(num_layers):
            attentions.append(_mk_transformer(attn_num_head_channels, in_channels, cross_attention_dim, resnet_groups,
                                              use_linear_projection, False, upcast_attention))
            motion_modules.append(_mk_motion(in_channels, use_motion_module, motion_module_type, motion_module_kwargs))
            resnets.append(res())
        self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)
        self.motion_modules = nn.ModuleList(motion_modules) if use_motion_module else motion_modules

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None,
                motion_module_alpha=1., cross_attention_kwargs=None, motion_cross_attention_kwargs=None):
        x4, b, f = _frames_first(hidden_states)
        temb_rep = None if self.resnets[0]._t_pre is not None else temb.repeat_interleave(f, dim=0)
        ls = getattr(self, "lora_scale", None)
        cross_kw = {"scale": ls} if ls is not None else cross_attention_kwargs
        motion_kw = self._motion_kwargs(motion_cross_attention_kwargs)
        x4 = self.resnets[0](x4, temb_rep)
        for i, attn in enumerate(self.attentions):
            x4 = attn(x4, encoder_hidden_states=encoder_hidden_states, cross_attention_kwargs=cross_kw).sample
            mm = self.motion_modules[i] if len(self.motion_modules) else None
            if mm is not None:
                x4 = _frames_first(mm(_frames_back(x4, b, f), encoder_hidden_states=encoder_hidden_states,
                                      cross_attention_kwargs=motion_kw))[0]
            x4 = self.resnets[i + 1](x4, temb_rep)
        return _frames_back(x4, b, f)


class _UpBlock(_Block3D):
    def _build(self, in_channels, out_channels, prev_output_channel, temb_channels, dropout, num_layers, resnet_eps,
               resnet_time_scale_shift, resnet_act_fn, resnet_groups, resnet_pre_norm, output_scale_factor,
               add_upsample, use_motion_module, motion_module_type, motion_module_kwargs, attn=None):
        resnets, attentions, motion_modules = [], [], []
        for i in range(num_layers):                        # unet_blocks.py:579-580 / 735-736
            res_skip_channels = in_channels if (i == num_layers - 1) else out_channels
            resnet_in_channels = prev_output_channel if i == 0 else out_channels
            resnets.append(_mk_resnet(resnet_in_channels + res_skip_channels, out_channels, temb_channels, resnet_eps,
                                      resnet_groups, resnet_act_fn, dropout, resnet_time_scale_shift,
                                      output_scale_factor, resnet_pre_norm))
            if attn is not None:
                attentions.append(_mk_transformer(attn["heads"], out_channels, attn["cross_dim"], resnet_groups,
                                                  attn["linear"], attn["only_cross"], attn["upcast"]))
            motion_modules.append(_mk_motion(out_channels, use_motion_module, motion_module_type, motion_module_kwargs))
        if attn is not None:
            self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)
        self.motion_modules = nn.ModuleList(motion_modules) if use_motion_module else motion_modules
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)]) \
            if add_upsample else None
        self.gradient_checkpointing = False

    def _up(self, hidden_states, res_hidden_states_tuple, temb, text, upsample_size, cross_kw, motion_kw):
        self._check_ckpt()
        x4, b, f = _frames_first(hidden_states)
        temb_rep = None if self.resnets[0]._t_pre is not None else temb.repeat_interleave(f, dim=0)
        for i in range(len(self.resnets)):
            skip4 = _frames_first(res_hidden_states_tuple[-1])[0]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            x4 = self._layer(i, x4, b, f, temb_rep, text, cross_kw, motion_kw, skip4)   # cat([x4, skip4]) read in place
        if self.upsamplers is not None:
            for up in self.upsamplers:
                x4 = up(x4, upsample_size)
        return _frames_back(x4, b, f)


class CrossAttnUpBlock3D(_UpBlock):
    """unet_blocks.py:543-706."""

    def __init__(self, in_channels: int, out_channels: int, prev_output_channel: int, temb_channels: int,
                 dropout: float = 0.0, num_layers: int = 1, resnet_eps: float = 1e-6,
                 resnet_time_scale_shift: str = "default", resnet_act_fn: str = "swish", resnet_groups: int = 32,
                 resnet_pre_norm: bool = True, attn_num_head_channels=1, cross_attention_dim=1280,
                 output_scale_factor=1.0, add_upsample=True, dual_cross_attention=False, use_linear_projection=False,
                 only_cross_attention=False, upcast_attention=False, use_motion_module=None, motion_module_type=None,
                 motion_module_kwargs=None):
        super().__init__()
        if dual_cross_attention:
            raise NotImplementedError
        self.has_cross_attention = True
        self.attn_num_head_channels = attn_num_head_channels
        self._build(in_channels, out_channels, prev_output_channel, temb_channels, dropout, num_layers, resnet_eps,
                    resnet_time_scale_shift, resnet_act_fn, resnet_groups, resnet_pre_norm, output_scale_factor,
                    add_upsample, use_motion_module, motion_module_type, motion_module_kwargs,
                    attn=dict(heads=attn_num_head_channels, cross_dim=cross_attention_dim, linear=use_linear_projection,
                              only_cross=only_cross_attention, upcast=upcast_attention))

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None,
                upsample_size=None, attention_mask=None, motion_module_alpha=1., cross_attention_kwargs=None,
                motion_cross_attention_kwargs={}):
        ls = getattr(self, "lora_scale", None)
        cross_kw = {"scale": ls} if ls is not None else cross_attention_kwargs
        return self._up(hidden_states, res_hidden_states_tuple, temb, encoder_hidden_states, upsample_size, cross_kw,
                        self._motion_kwargs(motion_cross_attention_kwargs))


class UpBlock3D(_UpBlock):
    """unet_blocks.py:709-817."""

    def __init__(self, in_channels: int, prev_output_channel: int, out_channels: int, temb_channels: int,
                 dropout: float = 0.0, num_layers: int = 1, resnet_eps: float = 1e-6,
                 resnet_time_scale_shift: str = "default", resnet_act_fn: str = "swish", resnet_groups: int = 32,
                 resnet_pre_norm: bool = True, output_scale_factor=1.0, add_upsample=True, use_motion_module=None,
                 motion_module_type=None, motion_module_kwargs=None):
        super().__init__()
        self._build(in_channels, out_channels, prev_output_channel, temb_channels, dropout, num_layers, resnet_eps,
                    resnet_time_scale_shift, resnet_act_fn, resnet_groups, resnet_pre_norm, output_scale_factor,
                    add_upsample, use_motion_module, motion_module_type, motion_module_kwargs)

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, upsample_size=None,
                encoder_hidden_states=None, motion_module_alpha=1., motion_cross_attention_kwargs={}, **kwargs):
        return self._up(hidden_states, res_hidden_states_tuple, temb, encoder_hidden_states, upsample_size, None,
                        self._motion_kwargs(motion_cross_attention_kwargs))
