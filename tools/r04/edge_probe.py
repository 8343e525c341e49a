"""The edge layers of cfg2 in the 16-bit modes through the c8 entry points (run under rocprofv3 --kernel-trace --stats for kernel-only times)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps  # noqa: E402
hip = RawOps("hip")
sp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x4 = torch.randn(1, 4, sp, sp, sp, device="cuda"); w4 = torch.randn(32, 4, 3, 3, 3, device="cuda") * 0.05
x32 = torch.relu(torch.randn(1, 32, sp, sp, sp, device="cuda")); w3 = torch.randn(3, 32, 3, 3, 3, device="cuda") * 0.05
b3 = torch.randn(3, device="cuda")
dy3 = torch.randn(1, 3, sp, sp, sp, device="cuda")
x4_16, x32_16, dy3_16 = hip.act16_pack(x4, 1), hip.act16_pack(x32, 1), hip.act16_pack(dy3, 1)
for _ in range(10):
    hip.conv3d_fwd_h16_c8(x4_16, 4, (sp, sp, sp), w4, compute=1, with_stats=True)        # d0.c0 forward (c4 kernel)
    hip.conv3d_bwd_data_h16_c8(dy3_16, 3, w3, (1, 32, sp, sp, sp), 1)                      # out conv data gradient (c4 kernel)
    hip.conv3d_fwd_h16(x32_16, 32, (sp, sp, sp), w3, bias=b3, compute=1, softmax=True)      # out conv forward + softmax
torch.cuda.synchronize()
