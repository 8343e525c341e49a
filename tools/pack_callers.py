#!/usr/bin/env python3
"""Who still packs fp32 -> c8 in a 16-bit train step (diagnostic; usage: pack_callers.py [bf16|fp16] [msseg2|dmri_hippo|cfg2]): call sites of ops.pack_act16 / _pack_scaled /
Act16.to_f32, counted over one step."""
import collections, os, sys, traceback, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import segmentation_pipeline_amd as sp
from segmentation_pipeline_amd import ops
from segmentation_pipeline_amd.models import ModularUNet, BlurConv3d, BlurConvTranspose3d
from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss
sp.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
arch = sys.argv[2] if len(sys.argv) > 2 else "msseg2"
if arch == "msseg2":
    m = ModularUNet(2, 2, [40, 40, 80, 80, 120, 120], 6, block_params={'residual': True}, downsample_class=BlurConv3d,
                    downsample_params={'kernel_size': 3, 'stride': 2, 'padding': 1}, upsample_class=BlurConvTranspose3d,
                    upsample_params={'kernel_size': 3, 'stride': 2, 'padding': 1, 'output_padding': 0}).cuda().train()
    shape = (1, 2, 96, 96, 96)
elif arch == "dmri_hippo":
    from segmentation_pipeline_amd.models import NestedResUNet
    m = NestedResUNet(3, 2, 40, dropout_p=0.2).cuda().train()
    shape = (8, 3, 48, 88, 24)
else:   # cfg2
    from functools import partial
    from torch import nn
    m = ModularUNet(4, 3, [32, 64, 128, 256, 320], 5, block_params={'normalization_class': partial(nn.GroupNorm, 8)},
                    upsample_class=nn.ConvTranspose3d, upsample_params={'kernel_size': 2, 'stride': 2}).cuda().train()
    shape = (1, 4, 128, 128, 128)
ncls = m.out_conv.out_channels
x = torch.randn(*shape, device="cuda")
y = torch.nn.functional.one_hot(torch.randint(0, ncls, (shape[0],) + shape[2:], device="cuda"), ncls).permute(0, 4, 1, 2, 3).float().contiguous()
cnt = collections.Counter()
def wrap(name, fn):
    def w(*a, **k):
        st = traceback.extract_stack(limit=7)[:-1]
        cnt[(name, " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in reversed(st[-5:])))] += 1
        return fn(*a, **k)
    return w
ops.pack_act16 = wrap("pack_act16", ops.pack_act16)
ops._pack_scaled = wrap("_pack_scaled", ops._pack_scaled)
ops._unpack_scaled = wrap("_unpack_scaled", ops._unpack_scaled)
ops.Act16.to_f32 = wrap("to_f32", ops.Act16.to_f32)
crit = HybridLogisticDiceLoss()
if "infer" in sys.argv:     # the no-grad c8 flow
    m.eval()
    with torch.no_grad():
        m(x)
else:
    crit(m(x), y)["loss"].backward()
torch.cuda.synchronize()
for (n, st), c in cnt.most_common(30):
    print(c, n, st)
