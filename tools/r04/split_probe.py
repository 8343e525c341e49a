#!/usr/bin/env python3
"""fp32 conv through bf16 MFMAs on an exact 3-way split (x = hi + mid + lo, each a bf16): six of the nine plane products
(the three dropped ones are below 2^-24 of the product) accumulated in the MFMA's fp32 accumulator.  Emulated here through
the EXISTING bf16 kernel by stacking the planes along the channel axis -- measures the accuracy the hardware delivers
against an fp64 reference, next to the native fp32 MFMA kernel."""
import os
import sys
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps  # noqa: E402


def split3(t, trunc):
    def cut(v):
        if trunc:
            return (v.view(torch.int32) & -65536).view(torch.float32)
        return v.to(torch.bfloat16).to(torch.float32)
    hi = cut(t)
    r1 = t - hi
    mid = cut(r1)
    r2 = r1 - mid
    lo = cut(r2)
    return hi, mid, lo, (r2 - lo)


def main():
    hip = RawOps("hip")
    torch.manual_seed(0)
    for ci, co, sp, relu in ((32, 32, 32, False), (32, 32, 32, True), (96, 32, 32, True), (256, 256, 16, True)):
        x = torch.randn(1, ci, sp, sp, sp)
        if relu:
            x = torch.relu(x)
        w = torch.randn(co, ci, 3, 3, 3) * (1.0 / (27 * ci) ** 0.5)
        ref = F.conv3d(x.double(), w.double(), padding=1)
        scale = ref.abs().max().item()
        y32 = hip.conv3d_fwd(x.cuda(), w.cuda()).cpu().double()
        ycpu = F.conv3d(x, w, padding=1).double()
        row = f"Cin {ci:3d} Cout {co:3d} {sp}^3 relu={int(relu)}: max|err|/max|y|  fp32 MFMA {((y32 - ref).abs().max() / scale):.2e}  torch CPU fp32 {((ycpu - ref).abs().max() / scale):.2e}"
        for trunc in (False, True):
            xh, xm, xl, xres = split3(x, trunc)
            wh, wm, wl, wres = split3(w, trunc)
            assert xres.abs().max() <= 2e-7 * x.abs().max() and wres.abs().max() <= 2e-7 * w.abs().max()
            for name, terms in (("x6", ((xh, wl), (xl, wh), (xm, wm), (xh, wm), (xm, wh), (xh, wh))),
                                ("x3", ((xh, wm), (xm, wh), (xh, wh))),
                                ("x9", ((xl, wl), (xm, wl), (xl, wm), (xh, wl), (xl, wh), (xm, wm), (xh, wm), (xm, wh), (xh, wh)))):
                xs = torch.cat([t[0] for t in terms], 1).cuda()
                ws_ = torch.cat([t[1] for t in terms], 1).cuda()
                x16 = hip.act16_pack(xs, 1)
                assert torch.equal(hip.act16_unpack(x16, xs.shape[1], (sp, sp, sp), 1), xs)   # the planes are bf16 values
                y = hip.conv3d_fwd_h16(x16, xs.shape[1], (sp, sp, sp), ws_, compute=1).cpu().double()
                row += f"  {name}{'t' if trunc else 'r'} {((y - ref).abs().max() / scale):.2e}"
        print(row, flush=True)


if __name__ == "__main__":
    main()
