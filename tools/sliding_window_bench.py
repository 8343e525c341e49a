#!/usr/bin/env python3
"""BASELINE cfg4: msseg2-style sliding-window inference at full size -- volume 4 x 256^3, patch 160, overlap 20,
overlap_mode 'average' (research/msseg2/msseg2.py:139-146) -> 8 patches of 4 x 160^3 per volume, cfg2 model with
2 outputs.  Single GPU here; with torch.distributed the tiles are sharded (tests/test_distributed_cpu.py).
Checks the size-independent property (probabilities of every voxel sum to 1, no voxel left uncovered)."""
import os
import sys
import time
from functools import partial

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmentation_pipeline_amd.models import ModularUNet  # noqa: E402
from segmentation_pipeline_amd.prediction import PatchPredict, grid_locations  # noqa: E402

torch.manual_seed(0)
model = ModularUNet(4, 2, [32, 64, 128, 256, 320], 5, block_params={'normalization_class': partial(nn.GroupNorm, 8)},
                    upsample_class=nn.ConvTranspose3d, upsample_params={'kernel_size': 2, 'stride': 2}).cuda().eval()
vol = torch.randn(4, 256, 256, 256, generator=torch.Generator().manual_seed(1234)).cuda()
pp = PatchPredict(patch_batch_size=1, patch_size=160, patch_overlap=20)
n = len(grid_locations(vol.shape[1:], (160,) * 3, (20,) * 3))
out = pp.predict_volume(model, vol)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 2
for _ in range(reps):
    out = pp.predict_volume(model, vol)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
s = out.sum(0)
assert out.shape == (2, 256, 256, 256) and torch.isfinite(out).all()
assert (s - 1).abs().max().item() < 1e-5, (s - 1).abs().max().item()
print(f"cfg4 sliding window: {n} patches of 4x160^3 per 4x256^3 volume, {dt * 1e3:.1f} ms per volume "
      f"({n / dt:.2f} patches/s), max |sum p - 1| = {(s - 1).abs().max().item():.1e}, "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
