"""Forward patches that inject the Object-Motion-Control features (`fmc/modified_modules.py`).

The trainers bind these onto the down blocks and tag each block with `traj_fea_idx`
(`train_cam_obj_ctrl.py:317-329`):

    m.forward = Adapted_CrossAttnDownBlock3D_forward.__get__(m, m.__class__); m.traj_fea_idx = i

Same names, signatures and behaviour here: `traj_features` is popped from the kwargs (it must never reach
the attention processors, reference :54-58 / :131-135), the stock block body runs, and
`hidden + traj_features[self.traj_fea_idx]` replaces the block's last residual before the downsampler
(reference :115-117 / :172-174) -- as one `fmc_feature_add_fwd` pass that touches only the conditioned half
under classifier-free guidance.  `Adapted_TemporalTransformerBlock_forward` / `unet3d_forward` of the
reference are MotionCtrl leftovers that nothing references and are not built.
"""
from __future__ import annotations


class UNet3DConditionOutput:
    def __init__(self, sample):
        self.sample = sample


def Adapted_CrossAttnDownBlock3D_forward(self, hidden_states, temb=None, encoder_hidden_states=None,
                                         attention_mask=None, motion_module_alpha=1., cross_attention_kwargs={},
                                         motion_cross_attention_kwargs={}):
    cross_kw = dict(cross_attention_kwargs or {})
    traj_features = cross_kw.pop("traj_features", None)
    if isinstance(cross_attention_kwargs, dict):
        cross_attention_kwargs.pop("traj_features", None)          # the reference pops from the caller's dict
    lora_scale = getattr(self, "lora_scale", None)
    if lora_scale is not None:
        cross_kw["scale"] = lora_scale
    return self._down(hidden_states, temb, encoder_hidden_states, cross_kw,
                      self._motion_kwargs(motion_cross_attention_kwargs), traj_features)


def Adapted_DownBlock3D_forward(self, hidden_states, temb=None, encoder_hidden_states=None, motion_module_alpha=1.,
                                motion_cross_attention_kwargs={}, **kwargs):
    traj_features = kwargs.pop("traj_features", None)
    return self._down(hidden_states, temb, encoder_hidden_states, None,
                      self._motion_kwargs(motion_cross_attention_kwargs), traj_features)


def patch_unet_for_omc(unet) -> None:
    """Exactly what `train_cam_obj_ctrl.py:317-329` does to `unet.down_blocks`."""
    idx = 0
    for _name, m in unet.down_blocks.named_modules():
        if m.__class__.__name__ == "CrossAttnDownBlock3D":
            m.forward = Adapted_CrossAttnDownBlock3D_forward.__get__(m, m.__class__)
            m.traj_fea_idx = idx
            idx += 1
        elif m.__class__.__name__ == "DownBlock3D":
            m.forward = Adapted_DownBlock3D_forward.__get__(m, m.__class__)
            m.traj_fea_idx = idx
            idx += 1
