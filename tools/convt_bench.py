#!/usr/bin/env python3
"""Per-level timing of the k=2,s=2 ConvTranspose3d kernels (HBM-bound) through the C ABI.
Reports the achieved fraction of HBM bandwidth on the algorithmic bytes (x + y once each).
usage: python tools/convt_bench.py [--f32x3]   (--f32x3: M355_COMPUTE_F32X3, the split kernels where they exist)"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps  # noqa: E402
from conv_bench import timeit  # noqa: E402

LEVELS = [("u0", 64, 64, 64), ("u1", 128, 128, 32), ("u2", 256, 256, 16), ("u3", 320, 320, 8)]  # name, Cin, Cout, in size


def main():
    hip = RawOps("hip")
    compute = 3 if "--f32x3" in sys.argv else 0
    print(f"{'level':6s} {'Cin':>4s} {'Cout':>4s} {'S':>4s} " + " ".join(f"{o + ' ms':>14s} {'GB/s':>7s}" for o in
                                                                      ("fwd", "bwd_data", "bwd_weight")))
    tot = [0.0, 0.0, 0.0]
    for name, ci, co, sp in LEVELS:
        x = torch.randn(1, ci, sp, sp, sp, device="cuda")
        w = torch.randn(ci, co, 2, 2, 2, device="cuda") * 0.05
        b = torch.randn(co, device="cuda")
        dy = torch.randn(1, co, 2 * sp, 2 * sp, 2 * sp, device="cuda")
        nbytes = 4.0 * (x.numel() + dy.numel())
        row = f"{name:6s} {ci:4d} {co:4d} {sp:4d} "
        for i, fn in enumerate((lambda: hip.convt_fwd(x, w, b, compute=compute), lambda: hip.convt_bwd_data(dy, w, x.shape),
                                lambda: hip.convt_bwd_weight(x, dy, 2))):
            ms = timeit(fn, 10)
            tot[i] += ms
            row += f"{ms:14.3f} {nbytes / ms / 1e6:7.0f} "
        print(row, flush=True)
    print("total  " + " " * 15 + " ".join(f"{t:14.3f} {'':7s}" for t in tot))


if __name__ == "__main__":
    main()
