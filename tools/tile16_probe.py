import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from raw_ops import RawOps
from segmentation_pipeline_amd import _lib
hip = RawOps("hip")
def t(fn, it=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for (ci, co, S) in [(40, 40, 96), (80, 40, 96), (40, 80, 96), (80, 80, 48), (160, 80, 48)]:
    x = torch.randn(1, ci, S, S, S, device="cuda"); w = torch.randn(co, ci, 3, 3, 3, device="cuda") * .05
    row = f"{ci}->{co}@{S}: "
    for env in ({}, {"M355_TILE16": "0"}, {"M355_CONV_NTW": "2"}, {"M355_CONV_NTW": "2", "M355_TILE16": "0"}):
        for k in ("M355_TILE16", "M355_CONV_NTW"): os.environ.pop(k, None)
        os.environ.update(env); _lib.reload_tuning()
        ms = t(lambda: hip.conv3d_fwd(x, w))
        row += f"{env or 'default'} {ms:.3f} ms ({2*27*ci*co*S**3/ms/1e9:.0f} TF) plan {hip.conv_plan((1,ci,S,S,S), co)} | "
    print(row, flush=True)
