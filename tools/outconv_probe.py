import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from raw_ops import RawOps
hip = RawOps("hip")
x = torch.randn(1, 32, 128, 128, 128, device="cuda"); w = torch.randn(3, 32, 3, 3, 3, device="cuda") * .05; b = torch.randn(3, device="cuda")
for _ in range(5): hip.conv3d_fwd(x, w, b, softmax=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): hip.conv3d_fwd(x, w, b, softmax=True)
e1.record(); torch.cuda.synchronize()
print("out conv + softmax", e0.elapsed_time(e1) / 10 * 1e3, "us")
