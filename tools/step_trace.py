#!/usr/bin/env python3
"""Per-step GPU intervals (one event per step boundary, no syncs) and host enqueue timestamps of the bench train
loop: finds host stalls (GC, allocator) or GPU slow-downs that a mean over K steps hides."""
import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import segmentation_pipeline_amd as sp
from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss
from segmentation_pipeline_amd.prediction import StandardPredict
from segmentation_pipeline_amd.trainer import train_step
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sp.set_precision(mode)
cfg = bench.WORKLOADS["cfg2"]
model = bench.build_model(cfg).cuda()
crit = HybridLogisticDiceLoss()
opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.95)
pred = StandardPredict(image_names=["X", "y"])
x, _, y = bench.synth((1, cfg[0]) + cfg[4], cfg[1], 1234, "cuda")
batch = {"X": x, "y": y}
for _ in range(3):
    train_step(model, crit, opt, pred, batch, "cuda")
torch.cuda.synchronize()
if "--freeze" in sys.argv:
    gc.collect(); gc.freeze()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
host = []
ev[0].record()
t0 = time.perf_counter()
for i in range(K):
    train_step(model, crit, opt, pred, batch, "cuda")
    ev[i + 1].record()
    host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) * 1e3
gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(K)]
print(f"{mode}: {K} steps in {tot:.1f} ms ({tot / K:.2f} ms/step); gc {gc.get_count()} stats {[g['collections'] for g in gc.get_stats()]}")
print("gpu interval per step:", " ".join(f"{g:.1f}" for g in gpu))
print("host enqueue done at :", " ".join(f"{h:.0f}" for h in host))
print("mem allocated MiB", torch.cuda.memory_allocated() >> 20, "reserved", torch.cuda.memory_reserved() >> 20,
      "retries", torch.cuda.memory_stats().get("num_alloc_retries", 0), "cudaMalloc calls", torch.cuda.memory_stats().get("num_device_alloc", 0))
