#!/usr/bin/env python3
"""M355_COMPUTE_F32X3 against an fp64 convolution, next to the fp32 MFMA kernel (forward, data gradient; several shapes)."""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps  # noqa: E402

hip = RawOps("hip")
torch.manual_seed(1)
for (n, ci, co, d, h, w) in ((1, 32, 32, 16, 16, 32), (2, 8, 40, 9, 7, 33), (1, 96, 32, 12, 16, 16), (1, 20, 64, 8, 24, 8),
                              (1, 64, 64, 32, 32, 32), (1, 256, 320, 8, 8, 8), (1, 33, 7, 5, 6, 40)):
    x = torch.relu(torch.randn(n, ci, d, h, w)); wt = torch.randn(co, ci, 3, 3, 3) / (27 * ci) ** 0.5
    b = torch.randn(co); dy = torch.randn(n, co, d, h, w)
    ref = F.conv3d(x.double(), wt.double(), b.double(), padding=1)
    refdx = F.conv_transpose3d(dy.double(), wt.double(), padding=1)
    out = {}
    for c in (0, 3):
        y = hip.conv3d_fwd(x.cuda(), wt.cuda(), b.cuda(), compute=c).cpu().double()
        dx = hip.conv3d_bwd_data(dy.cuda(), wt.cuda(), x.shape, compute=c).cpu().double()
        out[c] = ((y - ref).abs().max() / ref.abs().max(), (dx - refdx).abs().max() / refdx.abs().max())
    plan = hip.conv_plan(x.shape, co, compute=3)
    print(f"{(n, ci, co, d, h, w)} plan {plan}: fwd err fp32 {out[0][0]:.2e} x3 {out[3][0]:.2e} | bwd_data err fp32 {out[0][1]:.2e} x3 {out[3][1]:.2e}", flush=True)
    assert out[3][0] < 3e-6 and out[3][1] < 3e-6
# the k2 s2 conv-transpose forward on the split (convt_k2s2_fwd_x3_kernel)
for (n, ci, co, d, h, w) in ((1, 64, 64, 8, 8, 16), (2, 17, 5, 3, 5, 7), (1, 320, 320, 4, 4, 4), (1, 130, 70, 2, 9, 20)):
    x, wt, b = torch.randn(n, ci, d, h, w), torch.randn(ci, co, 2, 2, 2) / ci ** 0.5, torch.randn(co)
    ref = F.conv_transpose3d(x.double(), wt.double(), b.double(), stride=2)
    e = {c: float((hip.convt_fwd(x, wt, b, compute=c).cpu().double() - ref).abs().max() / ref.abs().max()) for c in (0, 3)}
    print(f"conv-transpose k2s2 {(n, ci, co, d, h, w)}: fwd err fp32 {e[0]:.2e} x3 {e[3]:.2e}", flush=True)
    assert e[3] < 3e-6
print("ok")
