#!/usr/bin/env python3
"""Where does a tile of the c8 weight-gradient kernel (conv3_bww_c8_kernel) spend its cycles?  Diagnostic build:
    python segmentation-pipeline_amd/build.py --stamps            (in the build container)
    M355_LIB_PATH=segmentation-pipeline_amd/libm355seg_dbg.so python tools/bww_stamps.py
Mean over workgroups of the per-tile cycles of wave 0: issue of the next tile's loads + MFMA loop, wait at the barrier
before the commit, the LDS commit (includes the vmcnt wait for the prefetched tile), the barrier after it."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps
hip = RawOps("hip"); L = hip.lib
buf = np.zeros((1024, 8), dtype=np.uint64)
for (ci, co, sp) in [(32, 32, 128), (96, 32, 128), (64, 64, 64), (192, 64, 64), (128, 128, 32)]:
    x16 = hip.act16_pack(torch.randn(1, ci, sp, sp, sp, device="cuda"), 1)
    dy16 = hip.act16_pack(torch.randn(1, co, sp, sp, sp, device="cuda"), 1)
    for _ in range(2):
        hip.conv3d_bwd_weight_c8(x16, dy16, ci, co, (sp, sp, sp), 1, with_bias=False)
    torch.cuda.synchronize()
    assert L.m355_debug_h16_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
    b = buf.astype(np.float64); n = b[:, 5]; ok = n > 0
    per = lambda k: (b[ok, k] / n[ok]).mean()
    print(f"{ci:4d}->{co:3d} @{sp}^3: workgroups {int(ok.sum()):4d} tiles/WG {n[ok].mean():5.1f}  per tile: loads+mfma {per(0):7.0f}  barrier1 {per(1):6.0f}  "
          f"commit {per(2):6.0f}  barrier2 {per(3):6.0f} | lifetime {b[ok, 6].mean():9.0f} cycles")
