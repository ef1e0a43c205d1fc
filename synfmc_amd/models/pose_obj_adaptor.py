"""`CamObjPoseAdaptor` (`fmc/models/pose_obj_adaptor.py:7-23`): camera encoder -> U-Net with OMC features."""
from __future__ import annotations

from torch import nn

from .pose_adaptor import features_to_video


class CamObjPoseAdaptor(nn.Module):
    def __init__(self, unet, pose_encoder):
        super().__init__()
        self.unet = unet
        self.pose_encoder = pose_encoder

    def forward(self, noisy_latents, timesteps, encoder_hidden_states, pose_embedding, traj_features):
        assert pose_embedding.ndim == 5
        bs = pose_embedding.shape[0]
        pose_embedding_features = features_to_video(self.pose_encoder(pose_embedding), bs)
        return self.unet(noisy_latents, timesteps, encoder_hidden_states,
                         pose_embedding_features=pose_embedding_features, traj_features=traj_features).sample
