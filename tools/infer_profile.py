#!/usr/bin/env python3
"""No-grad forwards of the cfg2 workload only (for rocprofv3 kernel statistics of the inference path).
usage: python tools/infer_profile.py [fp32|bf16|fp16] [steps]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import segmentation_pipeline_amd as sp  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
wl = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
cfg = bench.WORKLOADS[wl]
sp.set_precision(mode)
model = bench.build_model(cfg).cuda().eval()
x, _, _ = bench.synth((1, cfg[0]) + cfg[4], cfg[1], 1234, "cuda")
with torch.no_grad():
    for _ in range(3):
        model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model(x)
    torch.cuda.synchronize()
print(f"{wl} {mode} inference: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms / forward")
