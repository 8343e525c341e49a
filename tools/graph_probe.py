#!/usr/bin/env python3
"""Can the no-grad forward be captured into a hipGraph (torch.cuda.CUDAGraph) and replayed?  Prints eager vs
graph latency and the max difference.   usage: graph_probe.py [fp32|bf16|fp16]"""
import os, sys, time, torch
from functools import partial
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmentation_pipeline_amd.models import ModularUNet

import segmentation_pipeline_amd as sp
sp.set_precision(sys.argv[1] if len(sys.argv) > 1 else "fp32")
torch.manual_seed(0)
m = ModularUNet(4, 3, [32, 64, 128, 256, 320], 5, block_params={'normalization_class': partial(nn.GroupNorm, 8)},
                upsample_class=nn.ConvTranspose3d, upsample_params={'kernel_size': 2, 'stride': 2}).cuda().eval()
x = torch.randn(1, 4, 128, 128, 128, device="cuda")
with torch.no_grad():
    for _ in range(3):
        y_ref = m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        m(x)
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / 10
    xs = x.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            m(xs)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ys = m(xs)
    xs.copy_(x)
    g.replay()
    torch.cuda.synchronize()
    print("max diff", (ys - y_ref).abs().max().item())
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / 10
print(f"eager {t_eager * 1e3:.3f} ms, graph {t_graph * 1e3:.3f} ms")
