#!/usr/bin/env python3
"""Can a whole train step (forward + loss + backward + SGD) be captured into a hipGraph and replayed?  Prints eager vs
replay step time and whether the two loss trajectories are bit-identical.   usage: graph_train_probe.py [msseg2|cfg2] [precision]"""
import copy, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import segmentation_pipeline_amd as sp
from segmentation_pipeline_amd.models import ModularUNet, BlurConv3d, BlurConvTranspose3d
from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss

which = sys.argv[1] if len(sys.argv) > 1 else "msseg2"
sp.set_precision(sys.argv[2] if len(sys.argv) > 2 else "bf16")
torch.manual_seed(0)
if which == "msseg2":
    model = ModularUNet(2, 2, [40, 40, 80, 80, 120, 120], 6, block_params={'residual': True}, downsample_class=BlurConv3d,
                        downsample_params={'kernel_size': 3, 'stride': 2, 'padding': 1}, upsample_class=BlurConvTranspose3d,
                        upsample_params={'kernel_size': 3, 'stride': 2, 'padding': 1, 'output_padding': 0})
    shape, ncls, cw = (1, 2, 96, 96, 96), 2, [1, 100]
else:
    cfg = bench.WORKLOADS["cfg2"]
    model, shape, ncls, cw = bench.build_model(cfg), (1, cfg[0]) + cfg[4], cfg[1], None
model = model.cuda().train()
model_g = copy.deepcopy(model)
crit = HybridLogisticDiceLoss(logistic_class_weights=cw)
x = torch.randn(shape, device="cuda")
lab = torch.randint(0, ncls, (shape[0],) + tuple(shape[2:]), device="cuda")
y = torch.nn.functional.one_hot(lab, ncls).permute(0, 4, 1, 2, 3).float().contiguous()
STEPS = 10


def run_eager(m):
    opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.95)
    losses = []
    def step():
        opt.zero_grad(set_to_none=True)
        ld = crit(m(x), y)
        ld["loss"].backward()
        opt.step()
        return ld["loss"].detach()
    for _ in range(3):
        losses.append(step().clone())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(STEPS):
        losses.append(step().clone())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS, torch.stack(losses)


def run_graph(m):
    opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.95)
    losses = []
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):                                   # warm-up on a side stream (momentum buffers, caches)
            opt.zero_grad(set_to_none=True)
            ld = crit(m(x), y)
            ld["loss"].backward()
            opt.step()
            losses.append(ld["loss"].detach().clone())
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        ld = crit(m(x), y)
        ld["loss"].backward()
        opt.step()
    static_loss = ld["loss"].detach()
    # the capture itself does not execute: the graph holds step 4 onwards
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(STEPS):
        g.replay()
        losses.append(static_loss.clone())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS, torch.stack(losses)


te, le = run_eager(model)
tg, lg = run_graph(model_g)
print(f"{which}: eager {te * 1e3:.2f} ms/step, graph replay {tg * 1e3:.2f} ms/step; trajectories bit-identical: {torch.equal(le, lg)}"
      f" (final loss {le[-1].item():.6f} vs {lg[-1].item():.6f})")
