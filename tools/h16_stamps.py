#!/usr/bin/env python3
"""Where does a chunk of the 16-bit conv kernel spend its cycles?  Needs the diagnostic build:
    python segmentation-pipeline_amd/build.py --stamps            (in the build container)
    M355_LIB_PATH=segmentation-pipeline_amd/libm355seg_dbg.so python tools/h16_stamps.py
Prints, per layer, the mean over workgroups of the per-chunk cycles of wave 0: MFMA loop, wait at the barrier
before the commit, the LDS commit (includes the vmcnt wait for the prefetched chunk), the barrier after it, and
the per-item epilogue."""
import ctypes as C
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps  # noqa: E402
hip = RawOps("hip")
L = hip.lib
buf = np.zeros((1024, 8), dtype=np.uint64)
for (ci, co, sp) in [(96, 32, 128), (32, 32, 128), (192, 64, 64), (64, 64, 64), (64, 64, 128), (32, 32, 64), (96, 32, 64), (64, 32, 64), (32, 64, 64)]:
    x = torch.randn(1, ci, sp, sp, sp, device="cuda"); w = torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.05
    x16 = hip.act16_pack(x, 1)
    for _ in range(2):
        hip.conv3d_fwd_h16_c8(x16, ci, (sp, sp, sp), w, compute=1)
    torch.cuda.synchronize()
    assert L.m355_debug_h16_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
    b = buf[:512].astype(np.float64)
    n = b[:, 5]
    ok = n > 0
    per = lambda k: (b[ok, k] / n[ok]).mean()
    items = 128 ** 3 // 512 if sp == 128 else (sp ** 3 // 512) * ((co + 31) // 32)
    print(f"{ci:4d}->{co:3d} @{sp}^3: chunks/WG {n[ok].mean():5.1f}  per chunk: mfma {per(0):7.0f}  barrier1 {per(1):6.0f}  commit {per(2):6.0f}  "
          f"barrier2 {per(3):6.0f} | epilogue/chunk {per(4):6.0f} | lifetime {b[ok, 6].mean():9.0f} cycles (s_memtime ticks)")
