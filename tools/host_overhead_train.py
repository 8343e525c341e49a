#!/usr/bin/env python3
"""Host-side enqueue time of a TRAIN step vs its end-to-end time (per step, per precision mode), and the spread
over individual steps: is the step GPU-bound, and are there host stalls (allocator, GC)?"""
import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import segmentation_pipeline_amd as sp
from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss
cfg = bench.WORKLOADS["cfg2"]
for mode in sys.argv[1:] or ("fp32", "bf16"):
    sp.set_precision(mode)
    torch.manual_seed(0)
    model = bench.build_model(cfg).cuda()
    crit = HybridLogisticDiceLoss()
    opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.95)
    x, _, y = bench.synth((1, cfg[0]) + cfg[4], cfg[1], 1234, "cuda")

    def step():
        model.train(); ld = crit(model(x), y); opt.zero_grad(); ld["loss"].backward(); opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(20):
        t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{mode}: per step host enqueue min/median/max {min(host):.2f}/{sorted(host)[10]:.2f}/{max(host):.2f} ms, end-to-end "
          f"{min(total):.2f}/{sorted(total)[10]:.2f}/{max(total):.2f} ms; 20 steps back to back: enqueue {(t1 - t0) * 50:.2f} ms/step, "
          f"end-to-end {(t2 - t0) * 50:.2f} ms/step; gc counts {gc.get_count()}, alloc retries {torch.cuda.memory_stats().get('num_alloc_retries', 0)}", flush=True)
sp.set_precision("fp32")
