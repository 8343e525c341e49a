#!/usr/bin/env python3
"""Does an HBM-bound kernel overlap with the (one-workgroup-per-CU, MFMA-bound) bwd-weight kernel when the
two are launched on different HIP streams?  Prints the serial and the concurrent wall time."""
import os
import sys
import time
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps  # noqa: E402

hip = RawOps("hip")
x = torch.randn(1, 96, 128, 128, 128, device="cuda")
dy = torch.randn(1, 32, 128, 128, 128, device="cuda")
w = torch.randn(32, 96, 3, 3, 3, device="cuda") * 0.05
a = torch.randn(1, 32, 128, 128, 128, device="cuda")
gamma, beta = torch.ones(32, device="cuda"), torch.zeros(32, device="cuda")
mean, rstd = hip.norm_stats(a, 8)[:2]


def mfma_work(kind):
    if kind == "bww":
        hip.conv3d_bwd_weight(x, dy, 3, with_bias=False)
    else:
        hip.conv3d_fwd(x, w)


def hbm_work():
    for _ in range(4):
        hip.norm_act_fwd(a, mean, rstd, gamma, beta, 8, 1)


side = torch.cuda.Stream()
for kind in ("bww", "fwd"):
    for _ in range(2):
        mfma_work(kind); hbm_work()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); mfma_work(kind); torch.cuda.synchronize(); t_m = time.perf_counter() - t0
    t0 = time.perf_counter(); hbm_work(); torch.cuda.synchronize(); t_h = time.perf_counter() - t0
    t0 = time.perf_counter()
    mfma_work(kind)
    with torch.cuda.stream(side):
        hbm_work()
    torch.cuda.synchronize()
    t_c = time.perf_counter() - t0
    print(f"{kind}: mfma {t_m * 1e3:.3f} ms, hbm-bound x4 {t_h * 1e3:.3f} ms, serial {1e3 * (t_m + t_h):.3f} ms, "
          f"two streams {t_c * 1e3:.3f} ms")
