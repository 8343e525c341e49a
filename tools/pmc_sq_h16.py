#!/usr/bin/env python3
"""A few launches of the 16-bit conv kernel on cfg2 layers (c8 input), for SQ counter passes (tools/pmc_sq_h16.sh)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps  # noqa: E402
hip = RawOps("hip")
for (ci, co, sp) in [(96, 32, 128), (32, 32, 128), (192, 64, 64)]:
    x = torch.randn(1, ci, sp, sp, sp, device="cuda"); w = torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.05
    x16 = hip.act16_pack(x, 1)
    for _ in range(3):
        hip.conv3d_fwd_h16_c8(x16, ci, (sp, sp, sp), w, compute=1)
torch.cuda.synchronize()
