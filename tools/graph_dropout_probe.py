#!/usr/bin/env python3
"""Does a hipGraph-captured train step redraw its Dropout3d masks on every replay?  (torch draws them with the CUDA
generator, whose philox offset is advanced per replay when the generator is registered with the graph.)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import segmentation_pipeline_amd as sp
from segmentation_pipeline_amd.models import NestedResUNet
from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss
from segmentation_pipeline_amd.trainer import GraphedTrainStep
sp.set_precision(sys.argv[1] if len(sys.argv) > 1 else "fp32")
torch.manual_seed(0)
m = NestedResUNet(3, 2, 8, dropout_p=0.5).cuda()
opt = torch.optim.SGD(m.parameters(), lr=0.0)      # lr 0: the only thing that changes between replays is the mask
gs = GraphedTrainStep(m, HybridLogisticDiceLoss(), opt)
x = torch.randn(2, 3, 16, 16, 16, device="cuda")
y = torch.nn.functional.one_hot(torch.randint(0, 2, (2, 16, 16, 16), device="cuda"), 2).permute(0, 4, 1, 2, 3).float().contiguous()
losses = [float(gs({"X": x, "y": y})["loss"]) for _ in range(8)]
print("losses over replays (lr = 0):", ["%.6f" % v for v in losses])
print("distinct values among the replays:", len(set(losses[1:])))
