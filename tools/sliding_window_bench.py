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
# phase table (a device synchronisation per stamp: slower than the overlapped run above, but it separates the
# per-tile work that shards over GPUs -- tile_gather + model -- from the serial part every aggregating rank repeats:
# exchange (one gather) + accumulate + finalize)
import segmentation_pipeline_amd as sp  # noqa: E402
for mode in (sys.argv[1:] or ["fp32", "bf16"]):
    sp.set_precision(mode)
    timings = {}
    pt = PatchPredict(patch_batch_size=1, patch_size=160, patch_overlap=20, timings=timings)
    pt.predict_volume(model, vol)
    timings.clear()
    for _ in range(reps):
        pt.predict_volume(model, vol)
    tot = sum(timings.values()) / reps
    serial = sum(v for k, v in timings.items() if k in ("exchange", "accumulate", "finalize", "aggregate")) / reps
    print(f"  [{mode}] phases per volume (ms): " + "  ".join(f"{k} {v / reps * 1e3:.2f}" for k, v in timings.items()) +
          f"  | total {tot * 1e3:.1f}; serial (not sharded over GPUs) {serial * 1e3:.2f} ms = {serial / tot * 100:.1f} % "
          f"-> Amdahl bound at 8 GPUs: {tot / (serial + (tot - serial) / 8):.2f}x")
sp.set_precision("fp32")
