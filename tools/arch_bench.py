#!/usr/bin/env python3
"""Step time of the two architectures the reference actually trained.
usage: python tools/arch_bench.py [msseg2|dmri_hippo|all] [fp32|bf16|fp16]"""
import gc, os, sys, time, torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmentation_pipeline_amd.models import ModularUNet, NestedResUNet, BlurConv3d, BlurConvTranspose3d
from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss

def run(name, model, shape, ncls, cw=None, steps=10, warmup=3):
    model = model.cuda()
    crit = HybridLogisticDiceLoss(logistic_class_weights=cw)
    opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.95)
    x = torch.randn(shape, device="cuda")
    lab = torch.randint(0, ncls, (shape[0],) + tuple(shape[2:]), device="cuda")
    y = torch.nn.functional.one_hot(lab, ncls).permute(0, 4, 1, 2, 3).float().contiguous()
    def step():
        model.train(); ld = crit(model(x), y); opt.zero_grad(); ld["loss"].backward(); opt.step(); model.eval()
    for _ in range(warmup): step()
    torch.cuda.synchronize()
    gc.collect(); gc.freeze()   # keep a full collection (~90 ms of host stall) out of the timed steps, as bench.py does
    t0 = time.perf_counter()
    for _ in range(steps): step()
    t_host = (time.perf_counter() - t0) / steps   # enqueue only: the loop has not synchronised yet
    torch.cuda.synchronize(); t_train = (time.perf_counter() - t0) / steps
    with torch.no_grad():
        for _ in range(warmup): model(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): model(x)
        t_hinf = (time.perf_counter() - t0) / steps
        torch.cuda.synchronize(); t_inf = (time.perf_counter() - t0) / steps
    # the same iteration replayed from a hipGraph (trainer.GraphedTrainStep): where the eager step is host-bound
    from segmentation_pipeline_amd.trainer import GraphedTrainStep
    gs = GraphedTrainStep(model, crit, opt, warmup=1)
    batch = {"X": x, "y": y}
    gs(batch); gs(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): gs(batch)
    torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / steps
    print(f"{name}: train {t_train*1e3:.1f} ms/step ({shape[0]/t_train:.2f} patches/s), infer {t_inf*1e3:.1f} ms ({shape[0]/t_inf:.2f} patches/s)"
          f" [host enqueue {t_host*1e3:.1f} / {t_hinf*1e3:.1f} ms]; train step replayed from a hipGraph {t_graph*1e3:.1f} ms/step", flush=True)

only = sys.argv[1] if len(sys.argv) > 1 else ""   # "msseg2" / "dmri_hippo" / "all": run one of the two (profiling)
only = "" if only == "all" else only
if len(sys.argv) > 2:                              # precision mode: fp32 (default) | bf16 | fp16
    import segmentation_pipeline_amd as sp
    sp.set_precision(sys.argv[2])
    print(f"[precision {sys.argv[2]}]", flush=True)
torch.manual_seed(0)
if only in ("", "msseg2"):
  run("msseg2 ModularUNet(2,2,[40,40,80,80,120,120],6,residual,Blur) 1x2x96^3",
    ModularUNet(2, 2, [40, 40, 80, 80, 120, 120], 6, block_params={'residual': True}, downsample_class=BlurConv3d,
                downsample_params={'kernel_size': 3, 'stride': 2, 'padding': 1}, upsample_class=BlurConvTranspose3d,
                upsample_params={'kernel_size': 3, 'stride': 2, 'padding': 1, 'output_padding': 0}),
    (1, 2, 96, 96, 96), 2, [1, 100])
if only in ("", "dmri_hippo"):
  run("dmri_hippo NestedResUNet(3,2,40) 8x3x48x88x24 (sagittal split of 4x3x96x88x24)",
    NestedResUNet(3, 2, 40), (8, 3, 48, 88, 24), 2)
