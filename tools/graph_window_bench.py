#!/usr/bin/env python3
"""Sliding window with SMALL patches (launch-bound forwards): eager vs hipGraph replay (PatchPredict(graph=True)).
usage: python tools/graph_window_bench.py [patch] [batch]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import segmentation_pipeline_amd as sp
from segmentation_pipeline_amd.prediction import PatchPredict

patch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.manual_seed(0)
model = bench._unet(4, 2).cuda().eval() if hasattr(bench, "_unet") else bench.build_model(bench.WORKLOADS["cfg2"]).cuda().eval()
vol = torch.randn(4, 192, 192, 192)
for mode in ("fp32", "bf16"):
    with sp.precision(mode):
        res = {}
        for graph in (False, True):
            pp = PatchPredict(patch_batch_size=pb, patch_size=patch, patch_overlap=patch // 8, graph=graph)
            v = vol.cuda()
            out = pp.predict_volume(model, v)          # warm-up (+ capture)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                out = pp.predict_volume(model, v)
            torch.cuda.synchronize()
            res[graph] = ((time.perf_counter() - t0) / 3, out)
        from segmentation_pipeline_amd.prediction import grid_locations
        n = len(grid_locations((192,) * 3, (patch,) * 3, (patch // 8,) * 3))
        same = torch.equal(res[False][1], res[True][1])
        print(f"{mode}: 4x192^3, patch {patch}, overlap {patch // 8}, {n} tiles, batch {pb}: eager {res[False][0] * 1e3:.1f} ms "
              f"({res[False][0] / n * 1e3:.2f} ms/tile), graph replay {res[True][0] * 1e3:.1f} ms ({res[True][0] / n * 1e3:.2f} ms/tile), "
              f"bit-identical: {same}", flush=True)
