#!/usr/bin/env python3
"""Known-byte-count kernels for calibrating FETCH_SIZE / WRITE_SIZE on gfx950 (see
MI355X_MICROARCH.md §HBM): a 16-B/lane stream (copy_channels, float4) and a 4-B/lane stream
(add on an odd element count -> scalar path), each far larger than the 256 MiB Infinity Cache."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmentation_pipeline_amd import _lib  # noqa: E402

L = _lib.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
n4 = 160 * 1024 * 1024          # 640 MB per tensor, float4 path
a = torch.randn(n4, device="cuda")
b = torch.empty_like(a)
for _ in range(3):
    L.m355_copy_channels(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), 1, 1, n4, 0, 0, st)
n1 = n4 - 1                     # odd count -> dword loads/stores
c = torch.randn(n1, device="cuda")
d = torch.empty(n1, device="cuda")
for _ in range(3):
    L.m355_add(C.c_void_p(a.data_ptr()), C.c_void_p(c.data_ptr()), C.c_void_p(d.data_ptr()), n1, st)
torch.cuda.synchronize()
print("calib done", n4 * 4, n1 * 4)
