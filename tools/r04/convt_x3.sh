#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY' 2>&1 | grep -v amdgpu
import sys, torch
sys.path.insert(0, "tests")
from raw_ops import RawOps
hip, oracle = RawOps("hip"), RawOps("oracle")
import torch.nn.functional as F
torch.manual_seed(0)
for (n, ci, co, d, h, w) in ((1, 64, 64, 8, 8, 16), (2, 17, 5, 3, 5, 7), (1, 320, 320, 4, 4, 4), (1, 130, 70, 2, 9, 20), (1, 8, 8, 1, 1, 1)):
    x, wt, b = torch.randn(n, ci, d, h, w), torch.randn(ci, co, 2, 2, 2) * 0.1, torch.randn(co)
    ref = F.conv_transpose3d(x.double(), wt.double(), b.double(), stride=2)
    e = {}
    for c in (0, 3):
        y = hip.convt_fwd(x, wt, b, compute=c).cpu().double()
        e[c] = float((y - ref).abs().max() / ref.abs().max())
    print((n, ci, co, d, h, w), "err fp32 %.2e x3 %.2e" % (e[0], e[3]), "ok" if e[3] < 3e-6 else "BAD")
PY
python tools/convt_bench.py 2>&1 | grep -v amdgpu
python tools/convt_bench.py --f32x3 2>&1 | grep -v amdgpu
