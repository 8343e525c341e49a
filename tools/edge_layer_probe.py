import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from raw_ops import RawOps
from segmentation_pipeline_amd import _lib
hip = RawOps("hip")
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
S = 128
x = torch.randn(1, 4, S, S, S, device="cuda"); w = torch.randn(32, 4, 3, 3, 3, device="cuda") * .05
dy3 = torch.randn(1, 3, S, S, S, device="cuda"); w3 = torch.randn(3, 32, 3, 3, 3, device="cuda") * .05
for env in ({}, {"M355_CONV_PERSISTENT": "0"}, {"M355_CONV_NTW": "8"}, {"M355_CONV_NTW": "2"}, {"M355_CONV_NTW": "8", "M355_CONV_PERSISTENT": "0"}, {"M355_CONV_SLOTS": "256"}, {"M355_CONV_SLOTS": "1024"}):
    for k in ("M355_CONV_PERSISTENT", "M355_CONV_NTW", "M355_CONV_SLOTS"): os.environ.pop(k, None)
    os.environ.update(env); _lib.reload_tuning()
    a = t(lambda: hip.conv3d_fwd(x, w))
    b = t(lambda: hip.conv3d_bwd_data(dy3, w3, (1, 32, S, S, S)))
    print(f"{str(env):60s} d0.c0 fwd {a:.3f} ms plan {hip.conv_plan((1,4,S,S,S), 32)}  out-conv bwd-data {b:.3f} ms", flush=True)
