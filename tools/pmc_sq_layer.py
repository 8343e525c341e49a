#!/usr/bin/env python3
"""Forward and weight-gradient launches of the largest cfg2 layer (96 -> 32 @ 128^3), for SQ counter passes: the split
kernels (M355_COMPUTE_F32X3, what precision "fp32" runs) and the fp32 MFMA kernels, post-ReLU activations as in the net."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps  # noqa: E402
hip = RawOps("hip")
x = torch.relu(torch.randn(1, 96, 128, 128, 128, device="cuda")); w = torch.randn(32, 96, 3, 3, 3, device="cuda") * 0.05
dy = torch.randn(1, 32, 128, 128, 128, device="cuda")
for _ in range(3):
    for compute in (3, 0):
        hip.conv3d_fwd(x, w, compute=compute)
        hip.conv3d_bwd_weight(x, dy, 3, with_bias=False, compute=compute)
torch.cuda.synchronize()
