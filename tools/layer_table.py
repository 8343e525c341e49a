#!/usr/bin/env python3
"""Per-layer table of the conv launches (ops.CONV_PROFILE) of one train step of the two architectures the reference
trained (tools/arch_bench.py): launches grouped by (op, GFLOP, algorithmic MB, plan), mean ms, TFLOP/s.
usage: python tools/layer_table.py [msseg2|dmri_hippo|cfg2] [fp32|bf16|fp16]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmentation_pipeline_amd import ops  # noqa: E402
from segmentation_pipeline_amd.models import ModularUNet, NestedResUNet, BlurConv3d, BlurConvTranspose3d  # noqa: E402
from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "msseg2"
if len(sys.argv) > 2:
    ops.set_precision(sys.argv[2])
torch.manual_seed(0)
if which == "msseg2":
    model = ModularUNet(2, 2, [40, 40, 80, 80, 120, 120], 6, block_params={'residual': True}, downsample_class=BlurConv3d,
                        downsample_params={'kernel_size': 3, 'stride': 2, 'padding': 1}, upsample_class=BlurConvTranspose3d,
                        upsample_params={'kernel_size': 3, 'stride': 2, 'padding': 1, 'output_padding': 0})
    shape, ncls = (1, 2, 96, 96, 96), 2
elif which == "cfg2":
    import bench
    cfg = bench.WORKLOADS["cfg2"]
    model, shape, ncls = bench.build_model(cfg), (1, cfg[0]) + cfg[4], cfg[1]
else:
    model, shape, ncls = NestedResUNet(3, 2, 40), (8, 3, 48, 88, 24), 2
model = model.cuda()
crit = HybridLogisticDiceLoss()
opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.95)
x = torch.randn(shape, device="cuda")
lab = torch.randint(0, ncls, (shape[0],) + tuple(shape[2:]), device="cuda")
y = torch.nn.functional.one_hot(lab, ncls).permute(0, 4, 1, 2, 3).float().contiguous()


def step():
    model.train()
    ld = crit(model(x), y)
    opt.zero_grad()
    ld["loss"].backward()
    opt.step()


import gc  # noqa: E402
import statistics  # noqa: E402
for _ in range(4):
    step()
gc.collect()
gc.freeze()          # (a full collection of the cyclic GC stalls the launch thread ~90 ms: see bench.py)
STEPS = 5
ops.CONV_PROFILE = []
for _ in range(STEPS):
    step()
torch.cuda.synchronize()
prof, ops.CONV_PROFILE = ops.CONV_PROFILE, None
samples = {}
for (tag, flops, e0, e1, plan, nbytes) in prof:
    samples.setdefault((tag, round(flops / 1e9, 2), round(nbytes / 1e6, 1), plan), []).append(e0.elapsed_time(e1))
# per launch: the MEDIAN over the steps (an event pair also spans host-side stalls between the two records)
groups = {k: [statistics.median(v) * len(v), len(v)] for k, v in samples.items()}
tot = sum(g[0] for g in groups.values()) / STEPS
print(f"{which} [{ops.get_precision()}]: conv launches of one train step, {tot:.2f} ms (event-timed, includes packing / reductions)")
for (tag, gf, mb, plan), (ms, n) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
    print(f"  {tag:18s} {gf:8.2f} GF {mb:7.1f} MB plan {str(plan):18s} x{n // STEPS:<3d} {ms / n:7.3f} ms  {gf / (ms / n):7.1f} TF/s  "
          f"{ms / STEPS:7.2f} ms/step")
