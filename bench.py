#!/usr/bin/env python3
"""Benchmark of the hot path: BASELINE.json's metric on its cfg2 workload.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one iteration of the reference's training loop (segmentation_trainer.py:162-180:
train() -> forward -> criterion -> zero_grad -> backward -> optimizer.step -> eval()) on one
synthetic 1x4x128^3 patch per rank, with the 5-level GroupNorm/ConvTranspose U-Net
(18.08 M params, fp32).  `value` = patches/s over all ranks for the K timed train steps
(inputs resident in HBM).  The no-grad inference forward is timed the same way and
reported under "infer".  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
from functools import partial

import torch
import torch.distributed as dist
from torch import nn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "3D patches/sec (128³, 4ch) train+infer at 1/2/4/8 MI355X; Dice vs CPU ref"
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
WORKLOADS = {
    # name: (in_ch, out_ch, filters, depth, patch)
    "cfg2": (4, 3, [32, 64, 128, 256, 320], 5, (128, 128, 128)),
    "cfg2-64": (4, 3, [32, 64, 128, 256, 320], 5, (64, 64, 64)),     # quick functional check
    "cfg5": (3, 7, [32, 64, 128, 256, 320], 5, (32, 256, 256)),
}


def synth(shape, n_classes, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g)
    lab = torch.randint(0, n_classes, (shape[0],) + tuple(shape[2:]), generator=g)
    y = torch.nn.functional.one_hot(lab, n_classes).permute(0, 4, 1, 2, 3).float().contiguous()
    return x.to(device), lab.to(device), y.to(device)


def build_model(cfg):
    from segmentation_pipeline_amd.models import ModularUNet
    cin, cout, filters, depth, _ = cfg
    torch.manual_seed(0)
    return ModularUNet(cin, cout, filters, depth, block_params={'normalization_class': partial(nn.GroupNorm, 8)},
                       upsample_class=nn.ConvTranspose3d, upsample_params={'kernel_size': 2, 'stride': 2})


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of a kernel from the committed PMC passes of this same command
    (profiles/r01_pmc_traffic.json: separate FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 fetch correction)."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            data = json.load(f)["kernels"]
        for name, rec in data.items():
            if kernel_substr in name:
                return rec["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(cfg, batch):
    """Stock torch-CPU restatement of the reference path (oracle/torch_ref.py, pinned to the real
    reference by tests/golden) on a bounded sample: ONE no-grad forward + ONE train step of the
    same workload on the host cores."""
    from oracle import torch_ref as R
    cin, cout, filters, depth, patch = cfg
    torch.manual_seed(0)
    model = build_model(cfg)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    spec = R.UNetSpec(cin, cout, filters, depth, norm="group", groups=8, up="convT")
    g = torch.Generator().manual_seed(1234)
    x = torch.randn((batch, cin) + patch, generator=g)
    lab = torch.randint(0, cout, (batch,) + patch, generator=g)
    y = torch.nn.functional.one_hot(lab, cout).permute(0, 4, 1, 2, 3).float().contiguous()
    # the GPU box gives one GPU a share of 16 host cores (os.cpu_count() reports the whole host)
    cores = min(os.cpu_count() or 1, int(os.environ.get("M355_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-3, momentum=0.95)
    with torch.no_grad():
        t0 = time.time()
        p = R.unet_forward(sd, spec, x, training=False)
        t_inf = time.time() - t0
    t0 = time.time()
    p = R.unet_forward(sd, spec, x, training=True)
    ld = R.hybrid_logistic_dice_loss(p, y)
    opt.zero_grad()
    ld["loss"].backward()
    opt.step()
    t_train = time.time() - t0
    return {"value": batch / t_train, "unit": "patches/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 no-grad forward ({t_inf:.2f} s) + 1 train step ({t_train:.2f} s) of the same "
                      f"{batch}x{cin}x{'x'.join(map(str, patch))} workload, torch-CPU restatement of the reference",
            "infer_value": batch / t_inf, "dice_loss": float(ld["dice_loss"]), "loss": float(ld["loss"]),
            "_probs": p.detach()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=1, help="patches per rank per step")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "fp16"],
                    help="arithmetic of the 3x3x3 conv fwd / data gradient (default: exact fp32, the BASELINE cfg2 mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-infer", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_ddp = os.environ.get("M355_FORCE_DDP", "0") == "1"  # exercise the RCCL path with one rank
    if world > 1 or force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from segmentation_pipeline_amd import distributed as D
    from segmentation_pipeline_amd import ops
    from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss
    from segmentation_pipeline_amd.prediction import StandardPredict
    from segmentation_pipeline_amd.trainer import PhaseTimer, hard_dice_from_counts, train_step

    import segmentation_pipeline_amd as sp
    sp.set_precision(args.precision)
    cfg = WORKLOADS[args.workload]
    cin, cout, filters, depth, patch = cfg
    model = build_model(cfg).to(device)
    crit = HybridLogisticDiceLoss()
    opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.95)  # research/msseg2/msseg2.py:94
    runner = D.PatchParallel(model, force_collectives=force_ddp) if (world > 1 or force_ddp) else model
    predictor = StandardPredict(image_names=["X", "y"])
    x, lab, y = synth((args.batch, cin) + patch, cout, 1234 + rank, device)
    batch = {"X": x, "y": y}

    def barrier():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # ---- Dice vs CPU reference: first forward from the seed-0 weights, before any update ----
    model.eval()
    with torch.no_grad():
        p0 = model(x)
    am, counts = ops.argmax_confusion(p0, lab.to(torch.int32))
    gpu_dice0 = float(crit(p0, y)["dice_loss"])
    hard0 = hard_dice_from_counts(counts)[0].tolist()

    for _ in range(args.warmup):
        train_step(runner, crit, opt, predictor, batch, device)
    # ---- timed region: exactly K train steps ----
    ops.CONV_PROFILE = [] if rank == 0 else None
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss_dict, _ = train_step(runner, crit, opt, predictor, batch, device)
    barrier()
    elapsed = time.perf_counter() - t0
    prof, ops.CONV_PROFILE = ops.CONV_PROFILE, None
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    value = world * args.batch * args.steps / elapsed

    # per-phase breakdown (TorchTimer semantics: a sync per stamp), outside the timed region
    timer = PhaseTimer(device)
    for _ in range(2):
        train_step(runner, crit, opt, predictor, batch, device, timer)
    phases = {k: v / 2 * 1e3 for k, v in timer.timestamps.items()}

    # ---- inference: K no-grad forwards ----
    infer = None
    if not args.no_infer:
        model.eval()
        with torch.no_grad():
            for _ in range(max(1, args.warmup)):
                model(x)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                model(x)
            barrier()
            ti = time.perf_counter() - t0
        ti_t = torch.tensor([ti], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(ti_t, op=dist.ReduceOp.MAX)
        ti = float(ti_t.item())
        infer = {"value": world * args.batch * args.steps / ti, "unit": "patches/s", "ms_per_step": ti / args.steps * 1e3}

    if rank == 0:
        # ---- roofline of the dominant kernel: conv3_mfma_fwd_p_kernel<4,32> (fp32 MFMA implicit GEMM),
        # all launches of that variant in the timed region (forward convs and data gradients) ----
        # the kernel variant (NTW, GX) that accumulates the most time is the dominant kernel
        per_plan = {}
        for (tag, f, e0, e1, plan) in prof:
            if plan is not None and plan[0] in (1, 3) and plan[3] == 1:
                per_plan.setdefault((plan[0], plan[1], plan[2]), []).append((f, e0.elapsed_time(e1)))
        dom = max(per_plan, key=lambda p: sum(ms for _, ms in per_plan[p])) if per_plan else None
        sel = per_plan.get(dom, [])
        roofline = None
        if sel:
            kname = "conv3_mfma_fwd_p_kernel" if dom[0] == 3 else "conv3_mfma_fwd_kernel"  # 3: persistent variant
            tot_f, tot_ms = sum(f for f, _ in sel), sum(ms for _, ms in sel)
            ach = tot_f / (tot_ms * 1e-3) / 1e12
            roofline = {"bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": pmc_traffic(f"{kname}<{dom[1]}, {dom[2]}>"),
                        "kernel": f"{kname}<{dom[1]},{dom[2]}>", "launches": len(sel),
                        "avg_launch_ms": tot_ms / len(sel), "avg_gflop_per_launch": tot_f / len(sel) / 1e9}
        by_tag = {}
        for tag, f, e0, e1, plan in prof:
            a = by_tag.setdefault(tag, [0.0, 0.0, 0])
            a[0] += f
            a[1] += e0.elapsed_time(e1)
            a[2] += 1
        conv_summary = {k: {"tflops": v[0] / (v[1] * 1e-3) / 1e12, "ms_per_step": v[1] / args.steps, "launches_per_step": v[2] / args.steps}
                        for k, v in by_tag.items()}

        out = {
            "metric": METRIC, "value": value, "unit": "patches/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else f"{args.precision} operands / f32 accumulate (3x3x3 convs), f32 elsewhere",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: train step (fwd+loss+bwd+SGD) of ModularUNet({cin},{cout},"
                                   f"{str(filters).replace(' ', '')},{depth},GroupNorm(8),ConvTranspose3d k2s2) on {args.batch}x{cin}x{'x'.join(map(str, patch))} per GPU",
                       "global_batch": world * args.batch, "params": sum(p.numel() for p in model.parameters()),
                       "parallelism": f"patch-parallel dp{world}" if world > 1 else "single GPU",
                       "optimizer": "torch.optim.SGD(lr=1e-3, momentum=0.95)"},
            "infer": infer, "phases_ms": phases, "conv_kernels": conv_summary, "roofline": roofline,
            "final_loss": float(loss_dict["loss"].detach()),
            "dice": {"gpu_soft_dice_loss_step0": gpu_dice0, "gpu_hard_dice_step0": hard0},
        }
        if not args.no_cpu_baseline and world == 1:
            cb = cpu_baseline(cfg, args.batch)
            p_cpu = cb.pop("_probs")
            # Dice vs CPU ref on identical synthetic volume and identical seed-0 weights
            am_cpu = p_cpu.argmax(dim=1)
            from oracle import torch_ref as R
            hard_cpu = [r[4] for r in R.hard_dice_table(am_cpu[0], lab[0].cpu(), cout)]
            out["dice"].update({
                "cpu_soft_dice_loss_step0": cb["dice_loss"], "cpu_hard_dice_step0": hard_cpu,
                "max_abs_prob_diff_vs_cpu": float((p0.cpu() - p_cpu).abs().max()),
                "argmax_mismatch_voxels": int((am.cpu().long() != am_cpu).sum()),
            })
            out["cpu_baseline"] = cb
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
