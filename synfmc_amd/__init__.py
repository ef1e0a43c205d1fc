"""synfmc_amd -- the denoising hot path of FMC (FudanCVL/SynFMC) rebuilt for AMD MI355X (gfx950 / CDNA4).

The package mirrors the reference's `fmc` namespace for this path (`fmc.models.*`, `fmc.adapter`,
`fmc.modified_modules`, `fmc.util.get_traj_features_v2`, `fmc.data.dataset.ray_condition`,
`fmc.pipelines.*`); the arithmetic runs in hand-written HIP kernels behind the C ABI of
`include/fmc_hip.h` (`synfmc_amd/lib/libfmc_hip.so`).  `install_as_fmc()` overlays these modules on the reference's own
`fmc` package (or stands in for it when the reference is absent) so the trainers run unchanged.
"""
import os as _os
import sys as _sys

# MIOpen picks NHWC kernels for channels-last tensors only when asked to on some PyTorch-ROCm builds
_os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")

__version__ = "0.1.0"


# hot-path modules of the reference that this package replaces one for one (same names, classes, signatures, state-dict keys)
_REPLACED = ("models.unet", "models.unet_cam_obj", "models.unet_blocks", "models.motion_module",
             "models.attention_processor", "models.pose_adaptor", "models.pose_obj_adaptor", "models.resnet",
             "adapter", "modified_modules", "pipelines.pipeline_animation", "pipelines.pipeline_animation_cm_om")
# reference modules that stay the reference's own (datasets, pose math, logging / video I/O) and only get the hot-path
# FUNCTION patched in: (module, attribute, our module)
_PATCHED = (("data.dataset", "ray_condition", "data.dataset"), ("util", "get_traj_features_v2", "util"))


def _real_fmc_spec(reference_root):
    """The spec of a real `fmc` package on sys.path (the reference checkout), or None."""
    import importlib.util
    if reference_root:
        root = _os.path.abspath(reference_root)
        if not _os.path.isdir(_os.path.join(root, "fmc")):
            raise FileNotFoundError(f"{root} has no `fmc` package")
        if root not in _sys.path:
            _sys.path.insert(0, root)
    mod = _sys.modules.get("fmc")
    if mod is not None:
        return None if getattr(mod, "__name__", "") == __name__ else mod.__spec__
    try:
        return importlib.util.find_spec("fmc")
    except (ImportError, ValueError):
        return None


def install_as_fmc(reference_root=None) -> str:
    """Make the reference's trainers run on this package.  Returns the mode that was installed.

    * "overlay" -- a real `fmc` package is importable (or `reference_root` names the SynFMC checkout): it stays in place
      with everything this project does NOT rebuild (`fmc.utils.util`, `fmc.data.utils`, the dataset classes of
      `fmc.data.dataset`, ...), and only the hot-path modules are swapped: `fmc.models.*`, `fmc.adapter`,
      `fmc.modified_modules`, `fmc.pipelines.*` resolve to this package, `ray_condition` is patched INTO the real
      `fmc.data.dataset` and `get_traj_features_v2` into the real `fmc.util`.  Every `from fmc... import ...` line of
      `train_cam_ctrl.py` / `train_cam_obj_ctrl.py` keeps working (tests/test_cpu_misc.py runs them).  Nothing of the
      reference's hot-path modules is imported, so `diffusers` is needed only where the trainers themselves use it.
    * "standalone" -- no reference on the path: the package itself answers to the name `fmc` (hot-path modules only)."""
    import importlib
    ours = importlib.import_module(__name__)
    spec = _real_fmc_spec(reference_root)
    if spec is None:
        _sys.modules.setdefault("fmc", ours)
        for sub in _REPLACED + ("models", "util", "data", "data.dataset", "pipelines"):
            _sys.modules.setdefault(f"fmc.{sub}", importlib.import_module(f"{__name__}.{sub}"))
        return "standalone"
    real = importlib.import_module("fmc")
    for sub in _REPLACED:
        mod = importlib.import_module(f"{__name__}.{sub}")
        parent, _, leaf = f"fmc.{sub}".rpartition(".")
        parent_mod = importlib.import_module(parent)              # `fmc.models` / `fmc.pipelines`: the reference's own packages
        _sys.modules[f"fmc.{sub}"] = mod
        setattr(parent_mod, leaf, mod)
    for sub, attr, src in _PATCHED:
        try:
            target = importlib.import_module(f"fmc.{sub}")        # the reference's module (its own third-party imports apply)
        except ImportError as e:
            import warnings
            warnings.warn(f"install_as_fmc: fmc.{sub} of the reference does not import here ({e}); "
                          f"`{attr}` is available as {__name__}.{src}.{attr}")
            continue
        fn = getattr(importlib.import_module(f"{__name__}.{src}"), attr)
        setattr(target, f"_reference_{attr}", getattr(target, attr, None))
        setattr(target, attr, fn)
    real.__dict__["__synfmc_amd_overlay__"] = True
    return "overlay"
