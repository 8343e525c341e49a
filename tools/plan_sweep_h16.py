#!/usr/bin/env python3
"""(NTW, split-K) sweep of the 16-bit forward conv (c8 in, c8 out incl. the split-K reduction) for the cfg2 layers
against the planner's pick -- the one-shot kernel variant by default.  usage: python tools/plan_sweep_h16.py [bwd]"""
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
bwd = "bwd" in sys.argv
for name, ci, co, sp in CFG2:
    if ci <= 4 or co <= 4:
        continue
    kin, mout = (co, ci) if bwd else (ci, co)
    x16 = hip.act16_pack(torch.randn(1, kin, sp, sp, sp, device="cuda"), 1)
    w = torch.randn(mout, kin, 3, 3, 3, device="cuda") * 0.05
    run = lambda: hip.conv3d_fwd_h16_c8(x16, kin, (sp, sp, sp), w, compute=1)
    for k in ("M355_CONV_NTW", "M355_CONV_KSPLIT"):
        os.environ.pop(k, None)
    _reload()
    base = timeit(run, 8)
    plan = hip.conv_plan((1, kin, sp, sp, sp), mout, 1)
    res = []
    for ntw in (4, 2, 1):
        for ks in (1, 2, 3, 4, 6, 8):
            os.environ["M355_CONV_NTW"], os.environ["M355_CONV_KSPLIT"] = str(ntw), str(ks)
            _reload()
            if hip.conv_plan((1, kin, sp, sp, sp), mout, 1)[1:] != (ntw, plan[2], min(ks, max(1, (kin + 15) // 16))) and False:
                continue
            try:
                res.append((timeit(run, 8), ntw, hip.conv_plan((1, kin, sp, sp, sp), mout, 1)[3]))
            except Exception:  # noqa: BLE001
                pass
    res = sorted(set(res))[:5]
    print(f"{name:6s} kin={kin:4d} mout={mout:4d} S={sp:3d} model {base * 1e3:6.1f} us plan {plan} | best5 " +
          " ".join(f"({n},{k}):{t * 1e3:.1f}" for t, n, k in res), flush=True)
