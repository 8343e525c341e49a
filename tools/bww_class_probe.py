#!/usr/bin/env python3
"""Relative cost of the weight-gradient pair classes (full 32 x 32 pair on 32x32x2 MFMAs vs the 16-channel
sub-tile classes on 16x16x4): one pair per launch, same forced split count, 96^3 / 48^3 / 24^3 volumes.  The ratios
calibrate the per-class split counts of plan_bww (csrc/conv3d.hip)."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps  # noqa: E402
from segmentation_pipeline_amd import _lib  # noqa: E402

hip = RawOps("hip")


def t(ci, co, S, reps=5):
    x = torch.randn(1, ci, S, S, S, device="cuda")
    dy = torch.randn(1, co, S, S, S, device="cuda")
    for _ in range(2):
        hip.conv3d_bwd_weight(x, dy, 3, with_bias=False)
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.conv3d_bwd_weight(x, dy, 3, with_bias=False)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


for S, ns in ((96, 256), (48, 216), (24, 108)):
    os.environ["M355_BWW_NSPLIT"] = str(ns)
    _lib.reload_tuning()
    full = t(32, 32, S)
    row = [f"{S}^3, {ns} splits: full pair {full:.3f} ms"]
    for name, ci, co, mult in (("o32 x c16", 16, 32, 1), ("o16 x c32", 32, 16, 1), ("o16 x c16", 16, 16, 1)):
        v = t(ci, co, S)
        row.append(f"{name} {v:.3f} ms ({v / full:.2f})")
    print("; ".join(row), flush=True)
os.environ.pop("M355_BWW_NSPLIT")
_lib.reload_tuning()
for (ci, co, S) in ((40, 40, 96), (80, 40, 96), (80, 80, 48), (160, 80, 48), (40, 40, 48)):
    a = t(ci, co, S)
    os.environ["M355_TILE16"] = "0"
    _lib.reload_tuning()
    b = t(ci, co, S)
    os.environ.pop("M355_TILE16")
    _lib.reload_tuning()
    print(f"{ci}->{co} @{S}^3: classes {a:.3f} ms, padded pairs {b:.3f} ms", flush=True)
