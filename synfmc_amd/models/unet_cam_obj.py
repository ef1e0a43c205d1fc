"""`fmc.models.unet_cam_obj` surface: the reference keeps a byte-identical copy of the base U-Net here plus
`UNet3DConditionModelCamObjCond`; this build has one implementation in
`synfmc_amd.models.unet`."""
from .unet import (UNet3DConditionModel, UNet3DConditionModelCamObjCond, UNet3DConditionOutput)  # noqa: F401
