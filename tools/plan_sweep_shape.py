#!/usr/bin/env python3
"""(NTW, split-K, queue / one-shot) sweep of the fp32 forward conv kernel for arbitrary shapes (the non-cubic volumes and
40 / 80-wide layers of the reference's dmri_hippo and msseg2 nets), next to the planner's pick.
usage: python tools/plan_sweep_shape.py [N Cin Cout D H W]..."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps
from segmentation_pipeline_amd._lib import reload_tuning as _reload  # noqa: E402
from conv_bench import timeit  # noqa: E402

SHAPES = [(8, 80, 40, 48, 88, 24), (8, 40, 80, 48, 88, 24), (8, 40, 40, 48, 88, 24), (8, 160, 80, 24, 44, 12),
          (8, 80, 160, 24, 44, 12), (8, 80, 80, 24, 44, 12), (8, 240, 80, 24, 44, 12), (8, 320, 160, 12, 22, 6),
          (8, 160, 160, 12, 22, 6), (8, 120, 40, 48, 88, 24),
          (1, 80, 40, 96, 96, 96), (1, 40, 80, 96, 96, 96), (1, 40, 40, 96, 96, 96), (1, 160, 80, 48, 48, 48),
          (1, 80, 160, 48, 48, 48), (1, 80, 80, 48, 48, 48), (1, 40, 80, 48, 48, 48), (1, 80, 40, 48, 48, 48),
          (1, 200, 80, 24, 24, 24), (1, 80, 120, 24, 24, 24), (1, 120, 120, 24, 24, 24), (1, 120, 80, 24, 24, 24),
          (1, 240, 120, 12, 12, 12), (1, 120, 120, 12, 12, 12)]
KNOBS = ("M355_CONV_NTW", "M355_CONV_KSPLIT", "M355_CONV_PERSISTENT")


def main():
    hip = RawOps("hip")
    shapes = SHAPES
    if len(sys.argv) > 6:
        a = list(map(int, sys.argv[1:]))
        shapes = [tuple(a[i:i + 6]) for i in range(0, len(a), 6)]
    for (N, ci, co, D, H, W) in shapes:
        x = torch.randn(N, ci, D, H, W, device="cuda")
        w = torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.05
        flops = 2.0 * 27 * ci * co * N * D * H * W
        for k in KNOBS:
            os.environ.pop(k, None)
        _reload()
        plan = hip.conv_plan(x.shape, co)
        base = timeit(lambda: hip.conv3d_fwd(x, w), 5)
        res = []
        for ntw in (8, 4, 2, 1):
            for ks in (1, 2, 3, 4):
                for pers in (0, 2):
                    os.environ.update(M355_CONV_NTW=str(ntw), M355_CONV_KSPLIT=str(ks), M355_CONV_PERSISTENT=str(pers))
                    _reload()
                    try:
                        res.append((timeit(lambda: hip.conv3d_fwd(x, w), 5), ntw, ks, pers))
                    except Exception:  # noqa: BLE001 (a combination the library refuses)
                        pass
        for k in KNOBS:
            os.environ.pop(k, None)
        _reload()
        res.sort()
        best = " ".join(f"({n},{k},{'q' if p else '1'}):{t * 1e3:.0f}" for t, n, k, p in res[:5])
        print(f"N={N} {ci:3d}->{co:3d} {D}x{H}x{W}: planner {plan} {base * 1e3:6.0f} us ({flops / base / 1e9:5.1f} TF) | best5 (ntw,ks,mode):us {best}",
              flush=True)


main()
