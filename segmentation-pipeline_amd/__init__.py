"""MI355X-native hot path for efirdc/Segmentation-Pipeline (3D U-Net train / infer).

Sub-packages mirror the reference's plug-in surface: `models` (ModularUNet, Block3d,
NestedResUNet, Blur/WS convs, ensembles) and `criterions` (HybridLogisticDiceLoss).
Everything executes through libm355seg.so (include/m355seg.h); see DESIGN.md.
"""
from . import _lib, ops  # noqa: F401
from . import models, criterions  # noqa: F401

from .ops import get_precision, precision, set_precision  # noqa: F401

__all__ = ["models", "criterions", "ops", "precision", "set_precision", "get_precision"]
