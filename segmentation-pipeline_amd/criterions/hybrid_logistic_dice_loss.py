"""HybridLogisticDiceLoss on the fused HIP reduction kernels.

Mirror of segmentation_pipeline/criterions/hybrid_logistic_dice_loss.py:6-43
(reference): same constructor and the same dict of three scalars.  One pass over
(prediction, target) produces the four per-(n, c) sums, a finalize kernel turns them
into the scalars, and the backward is closed-form from the saved sums.
"""
import torch
from torch import nn

from .. import ops


class HybridLogisticDiceLoss(nn.Module):
    def __init__(self, dice_weight=0.5, logistic_class_weights=None, square_dice=True):
        super().__init__()
        self.dice_weight = dice_weight
        self.logistic_class_weights = logistic_class_weights
        self.square_dice = square_dice

    def forward(self, prediction, target):
        weights = None
        if self.logistic_class_weights is not None:
            weights = torch.tensor(self.logistic_class_weights, dtype=torch.float32, device=prediction.device)
        loss, dice_loss, logistic_loss = ops.hybrid_logistic_dice_loss(
            prediction, target, self.dice_weight, weights, self.square_dice)
        return {'loss': loss, 'dice_loss': dice_loss, 'logistic_loss': logistic_loss}
