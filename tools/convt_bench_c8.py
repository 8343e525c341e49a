#!/usr/bin/env python3
"""Per-level timing of the c8 conv-transpose kernels of the 16-bit training flow (forward, data gradient, weight
gradient on the 16-bit MFMA) through the C ABI, against the bytes each has to move.
usage: python tools/convt_bench_c8.py [workgroups-per-CU values ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conv_bench import timeit  # noqa: E402
from raw_ops import RawOps  # noqa: E402
from segmentation_pipeline_amd import _lib  # noqa: E402

LEVELS = [("u0", 64, 32, 64), ("u1", 128, 64, 32), ("u2", 256, 128, 16)]  # name, Cin, Cout, input size (cfg2 decoder)


def main():
    hip = RawOps("hip")
    for wgs in (sys.argv[1:] or ["0"]):
        os.environ["M355_CONVT_WGS"] = wgs
        _lib.reload_tuning()
        print(f"-- M355_CONVT_WGS={wgs}")
        for name, ci, co, sp in LEVELS:
            x = torch.randn(1, ci, sp, sp, sp, device="cuda")
            w = torch.randn(ci, co, 2, 2, 2, device="cuda") * 0.05
            b = torch.randn(co, device="cuda")
            dy = torch.randn(1, co, 2 * sp, 2 * sp, 2 * sp, device="cuda")
            x16, dy16 = hip.act16_pack(x, 1), hip.act16_pack(dy, 1)
            nbytes = 2.0 * (x.numel() + dy.numel())
            row = f"{name} {ci:4d}->{co:4d} @{sp:3d}^3  "
            for tag, fn in (("fwd", lambda: hip.conv_transpose3d_fwd_h16(x16, ci, (sp,) * 3, w, b, 1)),
                            ("bwd_data", lambda: hip.convt_bwd_data_h16(dy16, w, x.shape, 1)),
                            ("bwd_weight", lambda: hip.convt_bwd_weight_h16(x16, dy16, tuple(x.shape), co, 1))):
                ms = timeit(fn, 10)
                row += f"{tag} {ms * 1e3:7.1f} us ({nbytes / ms / 1e9:5.2f} TB/s)  "
            print(row, flush=True)


if __name__ == "__main__":
    main()
