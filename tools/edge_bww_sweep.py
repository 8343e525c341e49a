#!/usr/bin/env python3
"""Split-count sweep of the c8 weight-gradient kernel of the two EDGE layers (Cin <= 4 / Cout <= 4:
conv3_bww_c8_small_kernel) against the planner's pick; bytes / time = achieved HBM rate."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from raw_ops import RawOps
from segmentation_pipeline_amd._lib import reload_tuning as _reload
from conv_bench import timeit
hip = RawOps("hip")
for name, ci, co, sp in [("d0.c0", 4, 32, 128), ("out", 32, 3, 128), ("cfg5 d0.c0", 3, 32, (32, 256, 256)), ("cfg5 out", 32, 7, (32, 256, 256))]:
    sp3 = (sp,) * 3 if isinstance(sp, int) else sp
    x16 = hip.act16_pack(torch.randn(1, ci, *sp3, device="cuda"), 1)
    dy16 = hip.act16_pack(torch.randn(1, co, *sp3, device="cuda"), 1)
    run = lambda: hip.conv3d_bwd_weight_c8(x16, dy16, ci, co, sp3, 1)
    os.environ.pop("M355_BWW_NSPLIT", None); _reload()
    base = timeit(run, 8)
    nbytes = (x16.numel() + dy16.numel()) * 2
    res = []
    for ns in (128, 256, 384, 512, 768, 1024, 1280, 1536, 2048, 4096):
        os.environ["M355_BWW_NSPLIT"] = str(ns); _reload()
        res.append((timeit(run, 8), ns))
    print(f"{name:10s} {ci}->{co}: planner {base*1e3:6.1f} us ({nbytes/base/1e9:.2f} TB/s) | " + " ".join(f"{n}:{t*1e3:.0f}" for t, n in res), flush=True)
os.environ.pop("M355_BWW_NSPLIT", None)
