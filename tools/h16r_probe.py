#!/usr/bin/env python3
"""Timing of the register-weights 16-bit conv kernel on the cfg2 wide layers (c8 in, c8 out, weights pre-packed once per
call by the wrapper).  M355_H16R_DBG selects diagnostic loop variants (wrong results) of the diagnostic build:
    python segmentation-pipeline_amd/build.py --stamps
    M355_LIB_PATH=segmentation-pipeline_amd/libm355seg_dbg.so M355_H16R_DBG=11 python tools/h16r_probe.py"""
import os, sys, torch
os.environ.setdefault("M355_H16R", "1")     # (the kernel is opt-in; M355_H16R=0 times conv3_h16_kernel instead)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps
hip = RawOps("hip")
for name, ci, co, sp in [("d0.c1", 32, 32, 128), ("u0.c0", 96, 32, 128), ("d1.c1", 64, 64, 64), ("u1.c0", 192, 64, 64)]:
    x = torch.randn(1, ci, sp, sp, sp, device="cuda"); w = torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.05
    x16 = hip.act16_pack(x, 1)
    f = lambda: hip.conv3d_fwd_h16_c8(x16, ci, (sp, sp, sp), w, compute=1)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    msg = ""
    if int(os.environ.get("M355_H16R_DBG", "0")) & 8:
        y = f(); torch.cuda.synchronize()
        st = y.view(-1).view(torch.int64)[:256 * 4].view(256, 4).cpu().double()
        clk = (st[:, 0] / st[:, 1] * 100).median().item()
        msg = f" | in-kernel clock {clk:.0f} MHz, {(st[:, 0] / st[:, 2]).median().item():.0f} cycles/chunk incl. item overheads, {st[:, 2].median().item():.0f} chunks/WG"
    print(f"{name} {ci}->{co} @{sp}: {ms*1e3:.1f} us  {2*27*ci*co*sp**3/ms/1e9:.0f} TF (incl. weight pack + output alloc){msg}")
