"""Why does the 2nd replay of GraphedTrainStep diverge from the eager loop? (round-4 probe)"""
import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from functools import partial
import torch
from torch import nn
import segmentation_pipeline_amd as sp
from segmentation_pipeline_amd.models import ModularUNet
from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss
from segmentation_pipeline_amd.trainer import GraphedTrainStep

GN8 = {'normalization_class': partial(nn.GroupNorm, 8)}
CONVT = dict(upsample_class=nn.ConvTranspose3d, upsample_params={'kernel_size': 2, 'stride': 2})

def run(mode, interleave, warmup):
    torch.manual_seed(0)
    m_e = ModularUNet(4, 3, [8, 16], 2, block_params=dict(GN8), **CONVT).cuda().train()
    m_g = copy.deepcopy(m_e)
    g = torch.Generator().manual_seed(21)
    batches = []
    for _ in range(8):
        x = torch.randn((2, 4, 16, 16, 16), generator=g)
        lab = torch.randint(0, 3, (2, 16, 16, 16), generator=g)
        batches.append({"X": x.cuda(), "y": torch.nn.functional.one_hot(lab, 3).permute(0, 4, 1, 2, 3).float().contiguous().cuda()})
    crit = HybridLogisticDiceLoss()
    with sp.precision(mode):
        opt_e = torch.optim.SGD(m_e.parameters(), lr=1e-2, momentum=0.9)
        opt_g = torch.optim.SGD(m_g.parameters(), lr=1e-2, momentum=0.9)
        step = GraphedTrainStep(m_g, crit, opt_g, warmup=warmup)
        le, lg = [], []
        def e(b):
            opt_e.zero_grad(set_to_none=True)
            ld = crit(m_e(b["X"]), b["y"]); ld["loss"].backward(); opt_e.step()
            le.append(float(ld["loss"]))
        if interleave:
            for b in batches:
                e(b); lg.append(float(step(b)["loss"]))
        else:
            for b in batches: e(b)
            for b in batches: lg.append(float(step(b)["loss"]))
    print(mode, "interleave" if interleave else "separate", "warmup", warmup)
    print("  eager  ", " ".join(f"{v:.6f}" for v in le))
    print("  graphed", " ".join(f"{v:.6f}" for v in lg), "EQUAL" if le == lg else "DIFF")

print("env", {k: v for k, v in os.environ.items() if k.startswith("M355_")})
for mode in ("fp32",):
    run(mode, True, 1)
