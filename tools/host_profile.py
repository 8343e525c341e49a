#!/usr/bin/env python3
"""cProfile of the host side (Python + ctypes + torch allocator) of msseg2 steps in a 16-bit mode, where the step is
host-bound (~400 launches per train step): which functions the enqueue time goes to.
usage: python tools/host_profile.py [bf16|fp16|fp32] [train|infer]"""
import cProfile, io, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import segmentation_pipeline_amd as sp
from segmentation_pipeline_amd.models import ModularUNet, BlurConv3d, BlurConvTranspose3d
from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
what = sys.argv[2] if len(sys.argv) > 2 else "train"
sp.set_precision(mode)
torch.manual_seed(0)
model = ModularUNet(2, 2, [40, 40, 80, 80, 120, 120], 6, block_params={'residual': True}, downsample_class=BlurConv3d,
                    downsample_params={'kernel_size': 3, 'stride': 2, 'padding': 1}, upsample_class=BlurConvTranspose3d,
                    upsample_params={'kernel_size': 3, 'stride': 2, 'padding': 1, 'output_padding': 0}).cuda()
crit = HybridLogisticDiceLoss(logistic_class_weights=[1, 100])
opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.95)
x = torch.randn(1, 2, 96, 96, 96, device="cuda")
lab = torch.randint(0, 2, (1, 96, 96, 96), device="cuda")
y = torch.nn.functional.one_hot(lab, 2).permute(0, 4, 1, 2, 3).float().contiguous()


def train():
    model.train(); ld = crit(model(x), y); opt.zero_grad(); ld["loss"].backward(); opt.step()


def infer():
    with torch.no_grad():
        model(x)


fn = train if what == "train" else infer
if what == "infer":
    model.eval()
for _ in range(3):
    fn()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    fn()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue())
