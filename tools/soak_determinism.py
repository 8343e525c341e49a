#!/usr/bin/env python3
"""Soak test of the queue-driven persistent conv kernel and the swizzled weight-gradient kernel: the same
launch repeated many times must give bit-identical results every time (a race in the work queues, the LDS
hand-off of the next item, or the statistics partials would show up as a flipped bit sooner or later).
usage: python tools/soak_determinism.py [iterations]"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps
from segmentation_pipeline_amd._lib import reload_tuning as _reload  # the library caches the M355_* knobs  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
hip = RawOps("hip")
CASES = [  # (N, Cin, Cout, D, H, W, env)
    (1, 32, 32, 64, 64, 64, {}),
    (1, 96, 32, 64, 64, 64, {"M355_CONV_SLOTS": "37"}),
    (2, 24, 72, 20, 36, 40, {"M355_CONV_SLOTS": "11"}),
    (1, 8, 40, 16, 24, 32, {"M355_CONV_SLOTS": "5", "M355_CONV_KSPLIT": "2"}),
    (1, 4, 32, 32, 32, 32, {"M355_CONV_SLOTS": "9", "M355_CONV_PERSISTENT": "2"}),   # 2: queue-driven even for single-chunk items
]
for N, ci, co, D, H, W, env in CASES:
    for k in ("M355_CONV_SLOTS", "M355_CONV_KSPLIT", "M355_CONV_PERSISTENT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    _reload()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, ci, D, H, W, generator=g).cuda()
    w = (torch.randn(co, ci, 3, 3, 3, generator=g) * 0.1).cuda()
    b = torch.randn(co, generator=g).cuda()
    dy = torch.randn(N, co, D, H, W, generator=g).cuda()
    ref = None
    for i in range(iters):
        y = hip.conv3d_fwd(x, w, b)
        dx = hip.conv3d_bwd_data(dy, w, x.shape)
        dw, db = hip.conv3d_bwd_weight(x, dy, 3)
        st = hip.conv3d_fwd_stats(x, w, b, 0)
        cur = [y, dx, dw, db] + (list(st) if st is not None else [])
        if ref is None:
            ref = [t.clone() for t in cur]
        else:
            for j, (a, r) in enumerate(zip(cur, ref)):
                if not torch.equal(a, r):
                    print(f"NON-DETERMINISTIC: case {(N, ci, co, D, H, W, env)} output {j} iteration {i}: "
                          f"max diff {(a - r).abs().max().item():.3e}")
                    sys.exit(1)
    print(f"ok {iters} x {(N, ci, co, D, H, W)} {env}", flush=True)
# the c8 kernels of the 16-bit training flow: one-shot / queue-driven forward with fused statistics, data gradient, the
# ring-buffered + software-pipelined weight gradient
C8_CASES = [  # (N, Cin, Cout, D, H, W, env)
    (1, 32, 32, 32, 32, 64, {}),
    (1, 96, 32, 16, 20, 64, {"M355_BWW_NSPLIT": "7"}),
    (2, 24, 72, 9, 13, 33, {"M355_H16_ONESHOT": "3", "M355_CONV_SLOTS": "11"}),
    (1, 40, 64, 17, 12, 64, {}),
    (1, 64, 32, 8, 8, 96, {"M355_BWW_NSPLIT": "3"}),
]
KN = ("M355_CONV_SLOTS", "M355_CONV_KSPLIT", "M355_CONV_PERSISTENT", "M355_BWW_NSPLIT", "M355_H16_ONESHOT")
for N, ci, co, D, H, W, env in C8_CASES:
    for k in KN:
        os.environ.pop(k, None)
    os.environ.update(env)
    _reload()
    for compute in (1, 2):
        g = torch.Generator().manual_seed(2)
        x = torch.randn(N, ci, D, H, W, generator=g)
        w = (torch.randn(co, ci, 3, 3, 3, generator=g) * 0.1)
        b = torch.randn(co, generator=g)
        dy = torch.randn(N, co, D, H, W, generator=g)
        x16, dy16 = hip.act16_pack(x, compute), hip.act16_pack(dy, compute)
        ref = None
        for i in range(max(20, iters // 4)):
            y16, part = hip.conv3d_fwd_h16_c8(x16, ci, (D, H, W), w, b, compute=compute, with_stats=True)
            dx16 = hip.conv3d_bwd_data_h16_c8(dy16, co, w, (N, ci, D, H, W), compute)
            dw, db = hip.conv3d_bwd_weight_c8(x16, dy16, ci, co, (D, H, W), compute)
            up16 = hip.upsample_trilinear2x_bwd_h16(hip.upsample_trilinear2x_fwd_h16(x16, ci, (D, H, W), compute), ci, (D, H, W), compute)
            cur = [y16, part, dx16, dw, db, up16]
            if ref is None:
                ref = [t.clone() for t in cur]
            else:
                for j, (a, r) in enumerate(zip(cur, ref)):
                    if not torch.equal(a, r):
                        print(f"NON-DETERMINISTIC (c8): case {(N, ci, co, D, H, W, env)} compute {compute} output {j} iteration {i}")
                        sys.exit(1)
    print(f"ok c8 {(N, ci, co, D, H, W)} {env}", flush=True)
for k in KN:
    os.environ.pop(k, None)
_reload()
print("soak ok")
