"""`fmc/pipelines/pipeline_animation.py` surface: the reference file is the cm_om pipeline minus the OMC arguments
(9 small diff hunks).  `CameraCtrlPipeline` is the same loop without `traj_features`."""
from __future__ import annotations

from .pipeline_animation_cm_om import AnimationPipeline, AnimationPipelineOutput, CameraObjCtrlPipeline  # noqa: F401


class CameraCtrlPipeline(CameraObjCtrlPipeline):
    def __call__(self, prompt, pose_embedding, video_length, height=None, width=None, num_inference_steps: int = 50,
                 guidance_scale: float = 7.5, **kwargs):
        kwargs.pop("traj_features", None)
        return super().__call__(prompt, pose_embedding, video_length, traj_features=None, height=height, width=width,
                                num_inference_steps=num_inference_steps, guidance_scale=guidance_scale, **kwargs)
