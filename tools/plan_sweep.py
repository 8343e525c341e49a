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
from raw_ops import RawOps  # noqa: E402
from conv_bench import CFG2, timeit  # noqa: E402


def main():
    hip = RawOps("hip")
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
            base = timeit(lambda: hip.conv3d_fwd(x, w), 5)
            res = []
            for ntw in (8, 4, 2, 1):
                for ks in (1, 2, 3, 4, 6, 8):
                    if ks > kin // 4:
                        continue
                    os.environ["M355_CONV_NTW"], os.environ["M355_CONV_KSPLIT"] = str(ntw), str(ks)
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
