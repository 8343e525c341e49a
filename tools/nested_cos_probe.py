import sys, os, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import segmentation_pipeline_amd as sp
from segmentation_pipeline_amd import ops
from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss
from segmentation_pipeline_amd.models import NestedResUNet
g = np.load("tests/golden/nested_res_unet.npz")
sd = {k[5:]: torch.from_numpy(g[k]) for k in g.keys() if k.startswith("m.sd.")}
for mode in ("bf16", "fp16"):
    res = {}
    for c8 in (True, False):
        ops.H16_TRAIN_C8ONLY = c8
        m = NestedResUNet(3, 2, 8); m.load_state_dict(sd); m = m.cuda().train()
        with sp.precision(mode):
            p = m(torch.from_numpy(g["x"]).cuda())
            HybridLogisticDiceLoss()(p, torch.from_numpy(g["y"]).cuda())["loss"].backward()
        res[c8] = {k: v.grad.cpu().double().flatten() for k, v in m.named_parameters()}
        print(mode, c8, "probs err", (p.detach().cpu() - torch.from_numpy(g["m.probs_train"])).abs().max().item())
    for k in res[True]:
        ref = torch.from_numpy(g["m.grad." + k]).double().flatten()
        if ref.norm() < 1e-9: continue
        c = lambda a: float(a @ ref / (a.norm() * ref.norm() + 1e-300))
        print(f"{mode} {k:28s} c8 {c(res[True][k]):.4f} twin {c(res[False][k]):.4f} |ref| {ref.norm():.2e}")
