#!/usr/bin/env python3
"""Split-count sweep of the c8 weight-gradient kernel (m355_conv3d_bwd_weight_h16) for the cfg2 layers against the
planner's pick."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps  # noqa: E402
from segmentation_pipeline_amd._lib import reload_tuning as _reload  # noqa: E402
from conv_bench import CFG2, timeit  # noqa: E402

hip = RawOps("hip")
for name, ci, co, sp in CFG2:
    if ci <= 4 or co <= 4:
        continue
    x16 = hip.act16_pack(torch.randn(1, ci, sp, sp, sp, device="cuda"), 1)
    dy = torch.randn(1, co, sp, sp, sp, device="cuda")
    dy16 = hip.act16_pack(dy, 1)
    run = lambda: hip.conv3d_bwd_weight_h16(x16, dy16, dy, ci, co, (sp, sp, sp), 1, with_bias=False)
    os.environ.pop("M355_BWW_NSPLIT", None)
    _reload()
    base = timeit(run, 8)
    res = []
    for ns in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 170, 256, 384, 512, 768, 1024):
        os.environ["M355_BWW_NSPLIT"] = str(ns)
        _reload()
        res.append((timeit(run, 8), ns))
    res = sorted(res)[:5]
    print(f"{name:6s} Cin={ci:4d} Cout={co:4d} S={sp:3d} model {base * 1e3:6.1f} us | best5 nsplit:us " + " ".join(f"{n}:{t * 1e3:.1f}" for t, n in res), flush=True)
os.environ.pop("M355_BWW_NSPLIT", None)
