import os, sys, time, torch, collections
sys.path.insert(0, "/root/repo")
import segmentation_pipeline_amd as sp
from segmentation_pipeline_amd import ops
from segmentation_pipeline_amd.models import ModularUNet, BlurConv3d, BlurConvTranspose3d
from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss
sp.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
acc = collections.defaultdict(lambda: [0.0, 0])
def wrap(cls, name):
    f = getattr(cls, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); dt = time.perf_counter() - t0
        acc[cls.__name__ + "." + name][0] += dt; acc[cls.__name__ + "." + name][1] += 1
        return r
    setattr(cls, name, staticmethod(g))
for n in dir(ops):
    c = getattr(ops, n)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
        wrap(c, "forward"); wrap(c, "backward")
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
for _ in range(3): train()
torch.cuda.synchronize(); acc.clear()
t0 = time.perf_counter()
for _ in range(10): train()
th = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host enqueue {th*100:.2f} ms/step")
for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"  {k:34s} {t*100:7.3f} ms/step  {n/10:6.1f} calls/step  {t/n*1e6:6.1f} us/call")
