#!/usr/bin/env python3
"""conv3_bww_x3_kernel (M355_COMPUTE_F32X3 weight gradient) against an fp64 weight gradient, next to the fp32 MFMA kernel."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps  # noqa: E402

hip = RawOps("hip")
torch.manual_seed(2)
bad = 0
for (n, ci, co, d, h, w) in ((1, 32, 32, 6, 8, 32), (2, 8, 40, 9, 7, 33), (1, 96, 32, 12, 16, 16), (1, 20, 64, 8, 24, 8),
                              (1, 64, 64, 32, 32, 32), (1, 256, 320, 8, 8, 8), (1, 33, 7, 5, 6, 40), (3, 40, 40, 3, 10, 12),
                              (1, 32, 32, 1, 2, 32), (1, 16, 16, 17, 5, 70)):
    x = torch.relu(torch.randn(n, ci, d, h, w)); dy = torch.randn(n, co, d, h, w)
    xd = x.double().requires_grad_(False)
    wt = torch.zeros(co, ci, 3, 3, 3, dtype=torch.double, requires_grad=True)
    torch.nn.functional.conv3d(xd, wt, padding=1).backward(dy.double())
    ref = wt.grad
    out = {}
    for c in (0, 3):
        dw, db = hip.conv3d_bwd_weight(x, dy, 3, with_bias=True, compute=c)
        out[c] = float((dw.cpu().double() - ref).abs().max() / ref.abs().max())
        dbe = float((db.cpu().double() - dy.double().sum((0, 2, 3, 4))).abs().max())
    ok = out[3] < 3e-6
    bad += not ok
    print(f"{(n, ci, co, d, h, w)}: dw err fp32 {out[0]:.2e} x3 {out[3]:.2e} dbias {dbe:.1e} {'ok' if ok else 'BAD'}", flush=True)
print("ok" if not bad else f"{bad} BAD")
sys.exit(1 if bad else 0)
