#!/usr/bin/env python3
"""Host-side (Python + ctypes + launch) time per forward vs GPU time, fp32 and bf16 modes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import segmentation_pipeline_amd as sp
cfg = bench.WORKLOADS["cfg2"]
model = bench.build_model(cfg).cuda().eval()
x = torch.randn(1, 4, 128, 128, 128, device="cuda")
for mode in ("fp32", "bf16"):
    sp.set_precision(mode)
    with torch.no_grad():
        for _ in range(3): model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): model(x)
        t_host = (time.perf_counter() - t0) / 10      # enqueue only
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / 10
    print(f"{mode}: host enqueue {t_host*1e3:.2f} ms / forward, end-to-end {t_all*1e3:.2f} ms")
