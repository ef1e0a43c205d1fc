"""synfmc_amd -- the denoising hot path of FMC (FudanCVL/SynFMC) rebuilt for AMD MI355X (gfx950 / CDNA4).

The package mirrors the reference's `fmc` namespace for this path (`fmc.models.*`, `fmc.adapter`,
`fmc.modified_modules`, `fmc.util.get_traj_features_v2`, `fmc.data.dataset.ray_condition`,
`fmc.pipelines.*`); the arithmetic runs in hand-written HIP kernels behind the C ABI of
`include/fmc_hip.h` (`synfmc_amd/lib/libfmc_hip.so`).  `install_as_fmc()` registers the package under the
name `fmc` so the reference's trainers import it unchanged.
"""
import os as _os
import sys as _sys

# MIOpen picks NHWC kernels for channels-last tensors only when asked to on some PyTorch-ROCm builds
_os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")

__version__ = "0.1.0"


def install_as_fmc() -> None:
    """Make `import fmc...` resolve to this package (drop-in under train_cam_ctrl.py / train_cam_obj_ctrl.py)."""
    import importlib
    pkg = importlib.import_module(__name__)
    _sys.modules.setdefault("fmc", pkg)
    for sub in ("models", "models.unet", "models.unet_cam_obj", "models.unet_blocks", "models.motion_module",
                "models.attention_processor", "models.pose_adaptor", "models.pose_obj_adaptor", "models.resnet",
                "adapter", "modified_modules", "util", "data", "data.dataset", "pipelines",
                "pipelines.pipeline_animation", "pipelines.pipeline_animation_cm_om"):
        _sys.modules.setdefault(f"fmc.{sub}", importlib.import_module(f"{__name__}.{sub}"))
