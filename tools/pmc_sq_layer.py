#!/usr/bin/env python3
"""One forward and one weight-gradient launch of the largest cfg2 layer (96 -> 32 @ 128^3), for SQ counter passes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps  # noqa: E402
hip = RawOps("hip")
x = torch.randn(1, 96, 128, 128, 128, device="cuda"); w = torch.randn(32, 96, 3, 3, 3, device="cuda") * 0.05
dy = torch.randn(1, 32, 128, 128, 128, device="cuda")
for _ in range(3):
    hip.conv3d_fwd(x, w)
    hip.conv3d_bwd_weight(x, dy, 3, with_bias=False)
torch.cuda.synchronize()
