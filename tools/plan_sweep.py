#!/usr/bin/env python3
"""Times every (NTW, split-K) variant of the forward MFMA conv for the cfg2 layers (both directions) via
the M355_CONV_NTW / M355_CONV_KSPLIT overrides, next to what the built-in cost model picks.
usage: python tools/plan_sweep.py [--small]"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps
from segmentation_pipeline_amd._lib import reload_tuning as _reload  # the library caches the M355_* knobs  # noqa: E402
from conv_bench import CFG2, timeit  # noqa: E402


def sweep_bww(hip, layers):
    """weight gradient: every split count against the cost model's pick"""
    for name, ci, co, sp in layers:
        x = torch.randn(1, ci, sp, sp, sp, device="cuda")
        dy = torch.randn(1, co, sp, sp, sp, device="cuda")
        flops = 2.0 * 27 * ci * co * sp ** 3
        os.environ.pop("M355_BWW_NSPLIT", None)
        _reload()
        base = timeit(lambda: hip.conv3d_bwd_weight(x, dy, 3, with_bias=False), 5)
        res = []
        for ns in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 21, 24, 32, 42, 48, 64, 85, 96, 128, 170, 192, 256, 384, 512):
            os.environ["M355_BWW_NSPLIT"] = str(ns)
            _reload()
            res.append((timeit(lambda: hip.conv3d_bwd_weight(x, dy, 3, with_bias=False), 5), ns))
        os.environ.pop("M355_BWW_NSPLIT", None)
        _reload()
        res.sort()
        best = " ".join(f"{n}:{t * 1e3:.0f}" for t, n in res[:5])
        print(f"{name:6s} bww Cin={ci:4d} Cout={co:4d} S={sp:3d} model {base * 1e3:6.0f} us ({flops / base / 1e9:5.1f} TF) "
              f"| best5 nsplit:us {best}", flush=True)


def main():
    hip = RawOps("hip")
    if "--bww" in sys.argv:
        return sweep_bww(hip, [l for l in CFG2 if l[1] > 4 and l[2] > 4])
    layers = [l for l in CFG2 if l[1] > 4 and l[2] > 4]
    if "--small" in sys.argv:
        layers = [l for l in layers if l[3] <= 32]
    for name, ci, co, sp in layers:
        for direction in ("fwd", "bwd"):
            kin, mout = (ci, co) if direction == "fwd" else (co, ci)
            x = torch.randn(1, kin, sp, sp, sp, device="cuda")
            w = torch.randn(mout, kin, 3, 3, 3, device="cuda") * 0.05
            flops = 2.0 * 27 * kin * mout * sp ** 3
            for k in ("M355_CONV_NTW", "M355_CONV_KSPLIT"):
                os.environ.pop(k, None)
                _reload()
            base = timeit(lambda: hip.conv3d_fwd(x, w), 5)
            res = []
            for ntw in (8, 4, 2, 1):
                for ks in (1, 2, 3, 4, 6, 8):
                    if ks > kin // 4:
                        continue
                    os.environ["M355_CONV_NTW"], os.environ["M355_CONV_KSPLIT"] = str(ntw), str(ks)
                    _reload()
                    try:
                        res.append((timeit(lambda: hip.conv3d_fwd(x, w), 5), ntw, ks))
                    except RuntimeError:
                        pass
            res.sort()
            best = " ".join(f"({n},{k}):{t * 1e3:.0f}" for t, n, k in res[:5])
            print(f"{name:6s} {direction} kin={kin:4d} mout={mout:4d} S={sp:3d} model {base * 1e3:6.0f} us "
                  f"({flops / base / 1e9:5.1f} TF) | best5 {best}", flush=True)


if __name__ == "__main__":
    main()
