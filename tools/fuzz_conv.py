#!/usr/bin/env python3
"""Randomised parity sweep of the 3x3x3 conv family (forward / data gradient / weight gradient, fused
statistics) against the C oracle: random N, channels, ragged volumes, and random planner overrides so
that every kernel variant (one-shot / persistent, every NTW and lane-group width, split-K, both
bwd-weight generations) sees odd shapes.   usage: python tools/fuzz_conv.py [--convt] [cases] [seed]"""
import os
import random
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps
from segmentation_pipeline_amd._lib import reload_tuning as _reload  # the library caches the M355_* knobs  # noqa: E402


def rnd(*shape, seed=0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return (a - b).abs().max().item() / max(1.0, b.abs().max().item())


def fuzz_convt(cases, rng):
    """ConvTranspose3d k=2 s=2 (the MFMA GEMM kernels: every voxel-tile / channel-tile variant, fused bias
    gradient) on random channel counts and ragged volumes."""
    hip, oracle = RawOps("hip"), RawOps("oracle")
    worst = 0.0
    for i in range(cases):
        N = rng.choice([1, 1, 2])
        ci = rng.choice([1, 3, 4, 8, 17, 32, 40, 64, 65, 96, 130, 200, 330])
        co = rng.choice([1, 2, 4, 5, 16, 31, 32, 64, 70, 130])
        D, H, W = rng.randint(1, 9), rng.randint(1, 12), rng.randint(1, 20)
        x, w, b = rnd(N, ci, D, H, W, seed=3 * i), rnd(ci, co, 2, 2, 2, seed=3 * i + 1) * 0.2, rnd(co, seed=3 * i + 2)
        dy = rnd(N, co, 2 * D, 2 * H, 2 * W, seed=5 * i)
        e = [err(hip.convt_fwd(x, w, b), oracle.convt_fwd(x, w, b)),
             err(hip.convt_bwd_data(dy, w, x.shape), oracle.convt_bwd_data(dy, w, x.shape))]
        dwh, dbh = hip.convt_bwd_weight(x, dy, 2)
        dwo, dbo = oracle.convt_bwd_weight(x, dy, 2)
        e += [err(dwh, dwo), err(dbh, dbo)]
        worst = max(worst, max(e))
        if max(e) > 5e-5:
            print(f"MISMATCH convT case {i}: N={N} Cin={ci} Cout={co} DHW={D}x{H}x{W}", ["%.2e" % v for v in e], flush=True)
            sys.exit(1)
    print(f"convT fuzz ok: {cases} cases, worst relative error {worst:.2e}")


def main():
    if "--convt" in sys.argv:
        sys.argv.remove("--convt")
        return fuzz_convt(int(sys.argv[1]) if len(sys.argv) > 1 else 80,
                          random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0))
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    hip, oracle = RawOps("hip"), RawOps("oracle")
    worst = 0.0
    for i in range(cases):
        N = rng.choice([1, 1, 2, 3])
        ci, co = rng.choice([1, 2, 3, 4, 5, 8, 12, 17, 32, 40]), rng.choice([1, 3, 4, 7, 16, 31, 32, 33, 48, 70])
        D, H, W = rng.randint(1, 14), rng.randint(1, 22), rng.choice([1, 3, 4, 7, 8, 12, 16, 20, 24, 31, 32, 36, 40, 64])
        env = {}
        if rng.random() < 0.5:
            env["M355_CONV_SLOTS"] = str(rng.choice([1, 2, 3, 7, 16]))
        if rng.random() < 0.4:
            env["M355_CONV_NTW"] = str(rng.choice([1, 2, 4, 8]))
        if rng.random() < 0.4:
            env["M355_CONV_KSPLIT"] = str(rng.choice([1, 1, 2, 3]))
        if rng.random() < 0.2:
            env["M355_BWW_GEN"] = "1"
        for k in ("M355_CONV_SLOTS", "M355_CONV_NTW", "M355_CONV_KSPLIT", "M355_BWW_GEN"):
            os.environ.pop(k, None)
        os.environ.update(env)
        _reload()
        x, w, b = rnd(N, ci, D, H, W, seed=3 * i), rnd(co, ci, 3, 3, 3, seed=3 * i + 1) * 0.2, rnd(co, seed=3 * i + 2)
        add = rnd(N, co, D, H, W, seed=7 * i) if rng.random() < 0.3 else None
        dy = rnd(N, co, D, H, W, seed=5 * i)
        tag = f"case {i}: N={N} Cin={ci} Cout={co} DHW={D}x{H}x{W} env={env}"
        try:
            e = [err(hip.conv3d_fwd(x, w, b, add), oracle.conv3d_fwd(x, w, b, add)),
                 err(hip.conv3d_bwd_data(dy, w, x.shape), oracle.conv3d_bwd_data(dy, w, x.shape))]
            dwh, dbh = hip.conv3d_bwd_weight(x, dy, 3)
            dwo, dbo = oracle.conv3d_bwd_weight(x, dy, 3)
            e += [err(dwh, dwo), err(dbh, dbo)]
            st = hip.conv3d_fwd_stats(x, w, b, 0)
            if st is not None:
                so = oracle.conv3d_fwd_stats(x, w, b, 0)
                e += [err(st[0], so[0]), err(st[1], so[1]), err(st[2], so[2])]
        except Exception as ex:  # noqa: BLE001
            print("EXCEPTION", tag, repr(ex), flush=True)
            raise
        worst = max(worst, max(e))
        if max(e) > 5e-5:
            print("MISMATCH", tag, ["%.2e" % v for v in e], flush=True)
            sys.exit(1)
        if i % 20 == 0:
            print(f"{i} ok (worst so far {worst:.2e})", flush=True)
    print(f"fuzz ok: {cases} cases, worst relative error {worst:.2e}")


if __name__ == "__main__":
    main()
