"""The hot loop of SegmentationTrainer.train (segmentation_trainer.py:162-180, reference)
as a torchio-free harness: same order of operations and the same four TorchTimer phase
names (utils/torch_timer.py:6-30), so GPU and CPU-reference timings are comparable.
"""
import time
from typing import Dict, Optional

import torch

from . import distributed as D


class PhaseTimer:
    """TorchTimer semantics: a stream synchronisation before every stamp."""

    def __init__(self, device, enabled=True):
        self.device, self.enabled = device, enabled
        self.timestamps: Dict[str, float] = {}
        self.last = 0.0

    def start(self):
        if self.enabled and self.device.type != "cpu":
            torch.cuda.current_stream().synchronize()
        self.last = time.time()

    def stamp(self, name):
        if not self.enabled:
            return
        if self.device.type != "cpu":
            torch.cuda.current_stream().synchronize()
        now = time.time()
        self.timestamps[name] = self.timestamps.get(name, 0.0) + (now - self.last)
        self.last = now


def train_step(model, criterion, optimizer, predictor, batch, device, timer: Optional[PhaseTimer] = None):
    """One iteration: train() -> predict -> criterion -> zero_grad -> backward -> step -> eval().

    `model` may be a distributed.PatchParallel wrapper; gradient buckets are reduced
    between backward and the optimizer step.
    """
    if timer:
        timer.start()
        timer.stamp("data_loading")  # synthetic / resident data: ~0, kept for comparable reports
    model.train()
    batch = predictor.predict(model, device, batch)
    if timer:
        timer.stamp("model_forward")
    loss_dict = criterion(batch["y_pred"], batch["y"])
    if timer:
        timer.stamp("loss_function")
    if isinstance(model, D.PatchParallel):
        model.zero_grad()
    else:
        optimizer.zero_grad()
    loss_dict["loss"].backward()
    if isinstance(model, D.PatchParallel):
        model.finish_gradient_sync()
    optimizer.step()
    model.eval()
    if timer:
        timer.stamp("model_backward")
    return loss_dict, batch


def hard_dice_from_counts(counts: torch.Tensor) -> torch.Tensor:
    """dice = 2TP / (2TP + FP + FN) per (n, class) from ops.argmax_confusion's table
    (evaluators/segmentation_evaluator.py:74-86)."""
    tp, fp, fn = counts[..., 0].double(), counts[..., 1].double(), counts[..., 2].double()
    return 2 * tp / (2 * tp + fp + fn)
