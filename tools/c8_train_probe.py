#!/usr/bin/env python3
"""Round-3 probe (round 4: the fp16-rounded oracle carries the loss scale; the twin flow has none, its fp16 rows show why): the c8-only 16-bit TRAINING flow (ops.H16_TRAIN_C8ONLY) against the CPU oracle -- the reference's fp32
arithmetic and the rounding-matched variant (oracle.torch_ref.UNetSpec.rounding) -- and against the round-2 twin flow:
probabilities, loss, per-parameter gradient cosine / norm ratio, step time.   usage: c8_train_probe.py [small|cfg2|cfg5] ..."""
import os
import sys
import time
from functools import partial

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import segmentation_pipeline_amd as sp  # noqa: E402
from oracle import torch_ref as R  # noqa: E402
from segmentation_pipeline_amd import ops  # noqa: E402
from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss  # noqa: E402
from segmentation_pipeline_amd.models import ModularUNet  # noqa: E402

CASES = {"small": (4, 3, [16, 32, 64], 3, (1, 4, 32, 32, 32)),
         "cfg2": (4, 3, [32, 64, 128, 256, 320], 5, (1, 4, 128, 128, 128)),
         "cfg5": (3, 7, [32, 64, 128, 256, 320], 5, (1, 3, 32, 256, 256))}


def build(cin, cout, filters, depth):
    torch.manual_seed(0)
    return ModularUNet(cin, cout, filters, depth, block_params={'normalization_class': partial(nn.GroupNorm, 8)},
                       upsample_class=nn.ConvTranspose3d, upsample_params={'kernel_size': 2, 'stride': 2})


def oracle_run(sd0, spec, x, y, loss_scale=1.0):
    """loss_scale: the backward pass starts from loss * scale and the parameter gradients are divided by it -- what the
    fp16 mode's loss scaling does on the GPU (without it the fp16-rounded oracle's activation gradients underflow)"""
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd0.items()}
    p = R.unet_forward(sd, spec, x, training=True)
    ld = R.hybrid_logistic_dice_loss(p, y)
    (ld["loss"] * loss_scale).backward()
    return p.detach(), float(ld["loss"]), {k: v.grad / loss_scale for k, v in sd.items() if v.grad is not None}


def compare(tag, p, loss, grads, ref):
    p_ref, loss_ref, g_ref = ref
    cos, ratio, rel2 = [], [], []
    worst = None
    dot = na = nb = 0.0
    for k, g in grads.items():
        a, b = g.double().flatten(), g_ref[k].double().flatten()
        if not (torch.isfinite(a).all() and torch.isfinite(b).all()):
            print(f"    non-finite gradient: {k}")
            continue
        c = float((a @ b) / (a.norm() * b.norm() + 1e-300))
        cos.append(c)
        ratio.append(float(a.norm() / (b.norm() + 1e-300)))
        rel2.append(float((a - b).norm() / (b.norm() + 1e-300)))
        dot, na, nb = dot + float(a @ b), na + float(a @ a), nb + float(b @ b)
        if worst is None or c < worst[1]:
            worst = (k, c)
    print(f"  {tag:46s} max|dp| {float((p - p_ref).abs().max()):.2e} dloss {abs(loss - loss_ref):.1e} | per-parameter cosine min {min(cos):.5f} "
          f"({worst[0]}), norm ratio [{min(ratio):.4f}, {max(ratio):.4f}], rel L2 err max {max(rel2):.3f} median {sorted(rel2)[len(rel2) // 2]:.3f} | "
          f"all parameters: cosine {dot / (na * nb) ** 0.5:.6f}", flush=True)


def main():
    torch.set_num_threads(16)
    for name in (sys.argv[1:] or ["small"]):
        cin, cout, filters, depth, shape = CASES[name]
        model = build(cin, cout, filters, depth)
        sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        g = torch.Generator().manual_seed(1234)
        x = torch.randn(shape, generator=g)
        lab = torch.randint(0, cout, (shape[0],) + shape[2:], generator=g)
        y = torch.nn.functional.one_hot(lab, cout).permute(0, 4, 1, 2, 3).float().contiguous()
        print(f"== {name}: {shape}", flush=True)
        refs = {}
        for rounding in (None, "bf16", "fp16"):
            spec = R.UNetSpec(cin, cout, filters, depth, norm="group", groups=8, up="convT", rounding=rounding)
            t0 = time.time()
            scale = ops._auto_grad_scale(shape[0] * shape[2] * shape[3] * shape[4]) if rounding == "fp16" else 1.0
            refs[rounding] = oracle_run(sd0, spec, x, y, scale)
            print(f"  oracle rounding={rounding}: {time.time() - t0:.1f} s, loss {refs[rounding][1]:.6f}" +
                  (f" (loss scale 2^{int(scale).bit_length() - 1})" if scale != 1.0 else ""), flush=True)
        compare("oracle bf16-rounded vs fp32 oracle", refs["bf16"][0], refs["bf16"][1], refs["bf16"][2], refs[None])
        compare("oracle fp16-rounded vs fp32 oracle", refs["fp16"][0], refs["fp16"][1], refs["fp16"][2], refs[None])
        model = model.cuda().train()
        crit = HybridLogisticDiceLoss()
        xg, yg = x.cuda(), y.cuda()
        for mode in ("fp32", "bf16", "fp16"):
            for c8only in ((True,) if mode == "fp32" else (True, False)):
                ops.H16_TRAIN_C8ONLY = c8only
                with sp.precision(mode):
                    for rep in range(3):
                        model.zero_grad(set_to_none=True)
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        p = model(xg)
                        ld = crit(p, yg)
                        ld["loss"].backward()
                        torch.cuda.synchronize()
                        dt = time.perf_counter() - t0
                grads = {k: v.grad.detach().cpu() for k, v in model.named_parameters() if v.grad is not None}
                tag = f"{mode} {'c8-only' if c8only else 'twin'} ({dt * 1e3:.1f} ms fwd+bwd)"
                compare(tag + " vs fp32 oracle", p.detach().cpu(), float(ld["loss"]), grads, refs[None])
                if mode != "fp32":
                    compare(tag + f" vs {mode}-rounded oracle", p.detach().cpu(), float(ld["loss"]), grads, refs[mode])
        ops.H16_TRAIN_C8ONLY = True


if __name__ == "__main__":
    main()
