#!/usr/bin/env python3
"""Randomised parity sweep of the bandwidth-bound ops (norm statistics / norm+act forward and backward, pooling,
trilinear upsampling, softmax, loss, space-to-depth, conv-transpose) against the C oracle at shapes well beyond the
unit tests' -- errors that GROW with the extent (a coordinate rounded differently, a sum order) only show there.
usage: python tools/fuzz_misc.py [cases] [seed]"""
import os
import random
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps  # noqa: E402


def rnd(*shape, seed=0, scale=1.0, shift=0.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale + shift


def err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return (a - b).abs().max().item() / max(1.0, b.abs().max().item())


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    hip, oracle = RawOps("hip"), RawOps("oracle")
    worst = {}
    for i in range(cases):
        N, C_ = rng.choice([1, 1, 2, 3]), rng.choice([1, 2, 3, 4, 8, 12, 16, 24, 40])
        D, H, W = 2 * rng.randint(1, 12), 2 * rng.randint(1, 20), rng.choice([2, 4, 6, 10, 12, 22, 34, 40, 66, 70, 96])
        tag = f"case {i}: N={N} C={C_} DHW={D}x{H}x{W}"
        x = rnd(N, C_, D, H, W, seed=11 * i, scale=rng.choice([0.5, 1.0, 3.0]), shift=rng.choice([0.0, 0.0, 2.0, -5.0]))
        res = {}
        # normalisation: BatchNorm / InstanceNorm-like / GroupNorm, activation, residual add
        groups = rng.choice([0, 0] + [g for g in (1, 2, 4, 8, C_) if C_ % g == 0])
        act = rng.choice([0, 1, 2])
        gm, bt = rnd(C_, seed=i + 1) * 0.5 + 1, rnd(C_, seed=i + 2) * 0.3
        mean, rstd = hip.norm_stats(x, groups)[:2]
        mo, ro = oracle.norm_stats(x, groups)[:2]
        res["norm mean"], res["norm rstd"] = err(mean, mo), err(rstd, ro)
        add = rnd(N, C_, D, H, W, seed=i + 3) if rng.random() < 0.3 else None
        res["norm fwd"] = err(hip.norm_act_fwd(x, mo, ro, gm, bt, groups, act, add), oracle.norm_act_fwd(x, mo, ro, gm, bt, groups, act, add))
        dy = rnd(N, C_, D, H, W, seed=i + 4)
        gh, go = hip.norm_act_bwd(x, dy, mo, ro, gm, bt, groups, act), oracle.norm_act_bwd(x, dy, mo, ro, gm, bt, groups, act)
        res["norm dx"], res["norm dgamma"], res["norm dbeta"] = err(gh[0], go[0]), err(gh[1], go[1]), err(gh[2], go[2])
        # pooling / upsampling / space-to-depth
        po = oracle.avgpool_fwd(x)
        res["pool fwd"] = err(hip.avgpool_fwd(x), po)
        dp = rnd(*po.shape, seed=i + 5)
        res["pool bwd"] = err(hip.avgpool_bwd(dp, x.shape), oracle.avgpool_bwd(dp, x.shape))
        if D * H * W * C_ * N <= 400_000:
            uo = oracle.upsample_fwd(x)
            res["up fwd"] = err(hip.upsample_fwd(x), uo)
            du = rnd(*uo.shape, seed=i + 6)
            res["up bwd"] = err(hip.upsample_bwd(du, x.shape), oracle.upsample_bwd(du, x.shape))
            # the c8 kernels of the 16-bit flows: == the oracle on the rounded operand, rounded once (one 16-bit ulp)
            for compute, dt, ulp in ((1, torch.bfloat16, 2.0 ** -8), (2, torch.float16, 2.0 ** -11)):
                xr, dur = x.to(dt).float(), du.to(dt).float()
                y16 = hip.upsample_trilinear2x_fwd_h16(hip.act16_pack(x, compute), C_, (D, H, W), compute)
                ref = oracle.upsample_fwd(xr)
                got = y16.float().cpu().permute(0, 1, 3, 2).reshape(N, -1, 8 * D * H * W)[:, :C_].reshape(ref.shape)
                bad16 = ((got - ref).abs() > ulp * ref.abs() * 1.01 + 1e-5 * max(1.0, ref.abs().max().item())).sum().item()
                res[f"up fwd c8 ({'bf16' if compute == 1 else 'fp16'}) elements off"] = float(bad16)
                dx16 = hip.upsample_trilinear2x_bwd_h16(hip.act16_pack(du, compute), C_, (D, H, W), compute)
                refb = oracle.upsample_bwd(dur, x.shape)
                gotb = dx16.float().cpu().permute(0, 1, 3, 2).reshape(N, -1, D * H * W)[:, :C_].reshape(refb.shape)
                bad16 = ((gotb - refb).abs() > ulp * refb.abs() * 1.01 + 2e-5 * max(1.0, refb.abs().max().item())).sum().item()
                res[f"up bwd c8 ({'bf16' if compute == 1 else 'fp16'}) elements off"] = float(bad16)
        res["s2d"] = err(hip.space_to_depth(x), oracle.space_to_depth(x))
        # softmax + loss over the channels
        if C_ >= 2:
            so = oracle.softmax_fwd(x)
            res["softmax fwd"] = err(hip.softmax_fwd(x), so)
            res["softmax bwd"] = err(hip.softmax_bwd(so, dy), oracle.softmax_bwd(so, dy))
            lab = torch.randint(0, C_, (N, D, H, W), generator=torch.Generator().manual_seed(i))
            t = torch.nn.functional.one_hot(lab, C_).permute(0, 4, 1, 2, 3).float().contiguous()
            cw = (rnd(C_, seed=i + 7).abs() + 0.5) if rng.random() < 0.5 else None
            (o3h, sh), (o3o, so_) = hip.loss_fwd(so, t, 0.5, cw, True), oracle.loss_fwd(so, t, 0.5, cw, True)
            res["loss"] = err(o3h, o3o)
            res["loss bwd"] = err(hip.loss_bwd(so, t, so_, 1.0, 0.5, cw, True), oracle.loss_bwd(so, t, so_, 1.0, 0.5, cw, True))
        bad = {k: v for k, v in res.items() if v > 3e-5}
        for k, v in res.items():
            worst[k] = max(worst.get(k, 0.0), v)
        if bad:
            print("MISMATCH", tag, f"groups={groups} act={act}", {k: f"{v:.2e}" for k, v in bad.items()}, flush=True)
            sys.exit(1)
        if i % 10 == 0:
            print(f"{i} ok", flush=True)
    print(f"misc fuzz ok: {cases} cases; worst relative errors: " + ", ".join(f"{k} {v:.1e}" for k, v in sorted(worst.items())))


main()
