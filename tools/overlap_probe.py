#!/usr/bin/env python3
"""What happens to the weight-gradient kernel when some CUs are still held by another kernel (an RCCL gradient
bucket overlapping the backward pass) at the moment it starts?  tools/micro/cu_hog.hip holds `--hold` CUs for about
the kernel's own duration on a side stream; the weight gradient of the largest cfg2 layer is launched right behind
it on the main stream.  Static one-residency grid (M355_BWW_QUEUE=0): the workgroups mapped to the held CUs wait ->
~2x.  Queue-driven (default: 3 units per CU handed out by the dispatcher): the held CUs just take fewer units."""
import ctypes as C
import os
import sys
import time
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps  # noqa: E402
from segmentation_pipeline_amd import _lib  # noqa: E402

hog = C.CDLL(os.path.join(ROOT, "tools", "micro", "libcu_hog.so"))
hog.cu_hog.argtypes = [C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]
hip = RawOps("hip")
x = torch.randn(1, 96, 128, 128, 128, device="cuda")
dy = torch.randn(1, 32, 128, 128, 128, device="cuda")
sink = torch.zeros(4, device="cuda")
side = torch.cuda.Stream()
hold = int(sys.argv[sys.argv.index("--hold") + 1]) if "--hold" in sys.argv else 32


def run(with_hog, cycles):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if with_hog:
        hog.cu_hog(hold, cycles, C.c_void_p(sink.data_ptr()), C.c_void_p(side.cuda_stream))
        time.sleep(0.0002)  # let the hog workgroups land first
    e0.record()
    hip.conv3d_bwd_weight(x, dy, 3, with_bias=False)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for q in ("1", "0"):
    os.environ["M355_BWW_QUEUE"] = q
    _lib.reload_tuning()
    for _ in range(3):
        solo = run(False, 0)
    solo = min(run(False, 0) for _ in range(5))
    cycles = int(solo * 1e-3 * 2.0e9)          # hold the CUs for about one kernel duration (s_memtime ticks)
    held = min(run(True, cycles) for _ in range(5))
    print(f"M355_BWW_QUEUE={q}: weight gradient 96->32 @128^3 alone {solo:.3f} ms; with {hold} CUs held for ~{solo:.2f} ms "
          f"{held:.3f} ms ({held / solo:.2f}x)", flush=True)
