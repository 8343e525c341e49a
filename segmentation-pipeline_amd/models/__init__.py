"""Drop-in for `segmentation_pipeline.models` (reference models/__init__.py:1-4)."""
from .nested_residual_unet import NestedResUNet
from .components import WSConv3d, BlurConv3d, BlurConvTranspose3d, Block3d, StochasticMatrix
from .modular_unet import ModularUNet
from .ensemble import EnsembleOrientations, EnsembleModels, EnsembleFlips
