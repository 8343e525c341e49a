"""Drop-in for `segmentation_pipeline.criterions` (reference criterions/__init__.py:1)."""
from .hybrid_logistic_dice_loss import HybridLogisticDiceLoss
