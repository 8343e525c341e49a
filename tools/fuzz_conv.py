#!/usr/bin/env python3
"""Randomised parity sweep of the 3x3x3 conv family (forward / data gradient / weight gradient, fused
statistics) against the C oracle: random N, channels, ragged volumes, and random planner overrides so
that every kernel variant (one-shot / persistent, every NTW and lane-group width, split-K, both
bwd-weight generations) sees odd shapes.   usage: python tools/fuzz_conv.py [--convt | --h16 | --x3] [cases] [seed]
--x3: every qualifying layer on the split kernels (M355_F32X3=2: conv3_f32x3_kernel / conv3_bww_x3_kernel), same sweep."""
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


def fuzz_h16(cases, rng):
    """The c8 entry points of the 16-bit modes (forward with residual, c8-output forward, data gradient, weight
    gradient from c8 operands) on random shapes, both 16-bit types, and random planner overrides (4-wave / 8-wave
    variants, tiny residencies, split-K, tile heights) against the oracle on rounded operands."""
    hip, oracle = RawOps("hip"), RawOps("oracle")
    worst = 0.0
    knobs = ("M355_CONV_SLOTS", "M355_CONV_NTW", "M355_CONV_KSPLIT", "M355_H16_W8", "M355_H16_ONESHOT", "M355_BWW_NSPLIT", "M355_H16_ORDER")
    for i in range(cases):
        compute = rng.choice([1, 2])
        dt = torch.bfloat16 if compute == 1 else torch.float16
        N = rng.choice([1, 1, 2])
        ci, co = rng.choice([5, 8, 12, 17, 24, 32, 40, 72, 96]), rng.choice([5, 7, 8, 16, 31, 32, 33, 40, 64, 80])
        D, H, W = rng.randint(1, 17), rng.randint(1, 12), rng.choice([4, 8, 12, 16, 20, 31, 32, 33, 40, 62, 64])
        env = {}
        if rng.random() < 0.5:
            env["M355_CONV_SLOTS"] = str(rng.choice([1, 2, 3, 7, 16]))
        if rng.random() < 0.3:
            env["M355_CONV_NTW"] = str(rng.choice([1, 2, 4]))
        if rng.random() < 0.5:
            env["M355_CONV_KSPLIT"] = str(rng.choice([1, 1, 2, 3]))
        if rng.random() < 0.5:   # the queue-driven kernels (4-wave / 8-wave) instead of the default one-shot variant
            env["M355_H16_ONESHOT"] = "3"
            env["M355_H16_W8"] = str(rng.choice([0, 2, 2]))
        if rng.random() < 0.4:
            env["M355_BWW_NSPLIT"] = str(rng.choice([1, 2, 5]))
        if rng.random() < 0.4:
            env["M355_H16_ORDER"] = str(rng.choice([0, 1, 2]))
        for k in knobs:
            os.environ.pop(k, None)
        os.environ.update(env)
        _reload()
        x, w, b = rnd(N, ci, D, H, W, seed=3 * i), rnd(co, ci, 3, 3, 3, seed=3 * i + 1) * (1.0 / (27 * ci) ** 0.5), rnd(co, seed=3 * i + 2)
        add = rnd(N, co, D, H, W, seed=7 * i) if rng.random() < 0.3 else None
        dy = rnd(N, co, D, H, W, seed=5 * i)
        x16, dy16 = hip.act16_pack(x, compute), hip.act16_pack(dy, compute)
        tag = f"h16 case {i}: compute={compute} N={N} Cin={ci} Cout={co} DHW={D}x{H}x{W} env={env}"
        try:
            y = hip.conv3d_fwd_h16(x16, ci, (D, H, W), w, b, add, compute=compute)
            e = [err(y, oracle.conv3d_fwd(x, w, b, add, compute=compute)),
                 err(hip.conv3d_bwd_data_h16(dy16, co, w, x.shape, compute=compute), oracle.conv3d_bwd_data(dy, w, x.shape, compute=compute))]
            dwh, dbh = hip.conv3d_bwd_weight_h16(x16, dy16, dy, ci, co, (D, H, W), compute)
            dwo, dbo = oracle.conv3d_bwd_weight(x, dy, 3, compute=compute)
            e += [err(dwh, dwo), err(dbh, dbo)]
            if add is None:
                y16 = hip.conv3d_fwd_h16_c8(x16, ci, (D, H, W), w, b, compute=compute)
                got = y16.float().cpu().permute(0, 1, 3, 2).reshape(N, -1, D * H * W)[:, :co].reshape(N, co, D, H, W)
                yr = y.cpu()
                same = torch.equal(got, yr.to(dt).float())
                if not same:
                    print("C8 OUTPUT != ROUNDED FP32 OUTPUT", tag, flush=True)
                    sys.exit(1)
                dx16 = hip.conv3d_bwd_data_h16_c8(dy16, co, w, tuple(x.shape), compute)
                gx_ = dx16.float().cpu().permute(0, 1, 3, 2).reshape(N, -1, D * H * W)[:, :ci].reshape(N, ci, D, H, W)
                dxo = oracle.conv3d_bwd_data(dy, w, x.shape, compute=compute)
                if not bool(((gx_ - dxo).abs() <= ulp * dxo.abs() * 1.01 + 3e-5 * dxo.abs().max()).all()):
                    print("C8 DATA GRADIENT OFF", tag, flush=True)
                    sys.exit(1)
        except Exception as ex:  # noqa: BLE001
            print("EXCEPTION", tag, repr(ex), flush=True)
            raise
        worst = max(worst, max(e))
        if max(e) > 5e-5:
            print("MISMATCH", tag, ["%.2e" % v for v in e], flush=True)
            sys.exit(1)
        if i % 20 == 0:
            print(f"{i} ok (worst so far {worst:.2e})", flush=True)
    for k in knobs:
        os.environ.pop(k, None)
    _reload()
    print(f"h16 fuzz ok: {cases} cases, worst relative error {worst:.2e}")


def main():
    if "--h16" in sys.argv:
        sys.argv.remove("--h16")
        return fuzz_h16(int(sys.argv[1]) if len(sys.argv) > 1 else 100,
                        random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0))
    if "--convt" in sys.argv:
        sys.argv.remove("--convt")
        return fuzz_convt(int(sys.argv[1]) if len(sys.argv) > 1 else 80,
                          random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0))
    if "--x3" in sys.argv:
        sys.argv.remove("--x3")
        os.environ["M355_F32X3"] = "2"
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    hip, oracle = RawOps("hip"), RawOps("oracle")
    worst = 0.0
    for i in range(cases):
        N = rng.choice([1, 1, 2, 3])
        ci, co = rng.choice([1, 2, 3, 4, 5, 8, 12, 17, 32, 40, 80]), rng.choice([1, 3, 4, 7, 16, 31, 32, 33, 40, 48, 70, 80, 120])
        D, H, W = rng.choice([rng.randint(1, 14), 16, 32]), rng.choice([rng.randint(1, 22), 16, 32]), rng.choice([1, 3, 4, 7, 8, 12, 16, 20, 24, 31, 32, 36, 40, 64])
        env = {}
        if rng.random() < 0.5:
            env["M355_CONV_SLOTS"] = str(rng.choice([1, 2, 3, 7, 16]))
        if rng.random() < 0.4:
            env["M355_CONV_NTW"] = str(rng.choice([1, 2, 4, 8]))
        if rng.random() < 0.4:
            env["M355_CONV_KSPLIT"] = str(rng.choice([1, 1, 2, 3]))
        if rng.random() < 0.2:
            env["M355_BWW_GEN"] = "1"
        if rng.random() < 0.25:
            env["M355_TILE16"] = "0"
        if rng.random() < 0.4:   # item order: bit 0 = (y, z) tiles in 4x4 cubes, bit 1 = channel tile fastest
            env["M355_CONV_CUBE"] = str(rng.choice([0, 1, 2]))
        if rng.random() < 0.3:
            env["M355_CONV_PERSISTENT"] = "2"
        for k in ("M355_CONV_SLOTS", "M355_CONV_NTW", "M355_CONV_KSPLIT", "M355_BWW_GEN", "M355_TILE16", "M355_CONV_CUBE",
                  "M355_CONV_PERSISTENT"):
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
                e += [err(st[0], so[0]), err(st[1], so[1])]
                # rstd of a handful of samples is E[y^2] - E[y]^2 of fp32 sums: cancellation when they nearly coincide
                # (2 voxels per channel: 8e-5 seen); the normalisations of the models reduce over >= thousands
                if N * D * H * W >= 8:
                    e.append(err(st[2], so[2]) * (1.0 if N * D * H * W >= 64 else 0.2))
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
