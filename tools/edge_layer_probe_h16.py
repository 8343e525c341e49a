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
    return e0.elapsed_time(e1) / it * 1e3
S = 128
x4 = hip.act16_pack(torch.randn(1, 4, S, S, S, device="cuda"), 1); w4 = torch.randn(32, 4, 3, 3, 3, device="cuda") * .05
x32 = hip.act16_pack(torch.randn(1, 32, S, S, S, device="cuda"), 1); w3 = torch.randn(3, 32, 3, 3, 3, device="cuda") * .05
x32b = hip.act16_pack(torch.randn(1, 32, S, S, S, device="cuda"), 1); w32 = torch.randn(32, 32, 3, 3, 3, device="cuda") * .05
for env in ({}, {"M355_CONV_NTW": "2"}, {"M355_CONV_NTW": "1"}, {"M355_H16_ONESHOT": "3"}):
    for k in ("M355_H16_W8", "M355_CONV_NTW", "M355_CONV_SLOTS", "M355_H16_ONESHOT"): os.environ.pop(k, None)
    os.environ.update(env); _lib.reload_tuning()
    a = t(lambda: hip.conv3d_fwd_h16_c8(x4, 4, (S, S, S), w4, compute=1))
    b = t(lambda: hip.conv3d_fwd_h16(x32, 32, (S, S, S), w3, compute=1, softmax=True))
    c = t(lambda: hip.conv3d_fwd_h16_c8(x32b, 32, (S, S, S), w32, compute=1))
    print(f"32->32 (c8 out) {c:.1f} us plan {hip.conv_plan((1,32,S,S,S), 32, 1)} | ", end="")
    print(f"{str(env):60s} d0.c0 (c8 out) {a:.1f} us plan {hip.conv_plan((1,4,S,S,S), 32, 1)}  out conv + softmax {b:.1f} us plan {hip.conv_plan((1,32,S,S,S), 3, 1)}", flush=True)
