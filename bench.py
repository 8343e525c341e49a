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
import gc
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
# MI355X_MICROARCH.md: dense MFMA peaks (fp32: v_mfma_f32_32x32x2_f32; bf16 / fp16: v_mfma_f32_32x32x16_*), HBM3E
PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0, "fp16": 2500.0}
HBM_PEAK_GBS = 8000.0
# m355_conv3d_plan(): kernel family -> kernel name
PLAN_KERNEL = {1: "conv3_mfma_fwd_kernel", 3: "conv3_mfma_fwd_p_kernel", 2: "conv3_valu_smallcout_kernel",
               4: "conv3_h16_kernel", 5: "conv3_h16_kernel(8 waves)", 6: "conv3_h16_kernel(one-shot)", 0: "conv3d_direct_kernel"}
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


def source_hash():
    """sha1 over the kernel sources: PMC traffic recorded for other code is not quoted"""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "segmentation-pipeline_amd", "csrc")
    for name in sorted(os.listdir(d)):
        with open(os.path.join(d, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of a kernel from the PMC passes of this same command (tools/pmc_collect.sh ->
    profiles/r02_pmc_traffic.json: separate FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 fetch correction).  PMC
    counters cannot be read inside the run; the committed figure is only quoted when it was collected on
    exactly these kernel sources (source_hash), otherwise null."""
    path = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")
    try:
        with open(path) as f:
            doc = json.load(f)
        if doc.get("source_hash") != source_hash():
            return None
        # "conv3_mfma_fwd_p_kernel<4, 32>" also has to find "...<4, 32, false>" (trailing template arguments)
        stem = kernel_substr[:-1] if kernel_substr.endswith(">") else kernel_substr
        for name, rec in doc["kernels"].items():
            if kernel_substr in name or (stem + ",") in name:
                return rec["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def kernel_groups(prof, precision):
    """ops.CONV_PROFILE entries grouped by (kernel name, tile variant) -> [flops, bytes, ms, launches, {(tag, plan)}]"""
    groups = {}
    for (tag, flops, e0, e1, plan, nbytes) in prof:
        if tag == "conv3d_bwd_weight":
            key = ("conv3_mfma_bww2_kernel" if precision == "fp32" else "conv3_bww_c8_kernel", "")
        elif plan is None:
            continue
        else:
            key = (PLAN_KERNEL.get(plan[0], "conv3d"), f"<{plan[1]},{plan[2]}>" + (f" split-K {plan[3]}" if plan[3] > 1 else ""))
        g = groups.setdefault(key, [0.0, 0.0, 0.0, 0, set()])
        g[0] += flops
        g[1] += nbytes
        g[2] += e0.elapsed_time(e1)
        g[3] += 1
        g[4].add((tag, plan))
    return groups


def dominant_keys(prof, precision):
    """(tag, plan) pairs of the launches of the kernel that accumulates the most time, and a sampling stride: at most
    ~12 event-timed launches per step (a stride coprime with the launch count walks through all layers over the steps)"""
    import math
    groups = kernel_groups(prof, precision)
    if not groups:
        return None, 1
    g = groups[max(groups, key=lambda k: groups[k][2])]
    n, every = g[3], max(1, -(-g[3] // 12))
    while every > 1 and math.gcd(every, n) != 1:
        every += 1
    return g[4], every


def conv_summary_of(prof, steps):
    by_tag = {}
    for tag, f, e0, e1, plan, _nb in prof:
        a = by_tag.setdefault(tag, [0.0, 0.0, 0])
        a[0] += f
        a[1] += e0.elapsed_time(e1)
        a[2] += 1
    return {k: {"tflops": v[0] / (v[1] * 1e-3) / 1e12, "ms_per_step": v[1] / steps, "launches_per_step": v[2] / steps}
            for k, v in by_tag.items()}


def roofline_of(prof, precision, full_prof=None):
    """Roofline object of the dominant conv kernel of a timed region.  prof: ops.CONV_PROFILE entries
    (tag, flops, e0, e1, plan, bytes).  Kernels are grouped by (kernel name, tile variant); the group with
    the most accumulated time is the dominant kernel.  bound = whichever of algorithmic-bytes / 8 TB/s and
    flops / dense-MFMA-peak is larger over the group's launches; achieved / peak are reported in that unit."""
    groups = kernel_groups(prof, precision)
    if not groups:
        return None
    key = max(groups, key=lambda k: groups[k][2])
    flops, nbytes, ms, n = groups[key][:4]
    # share of the conv time: from the launch-by-launch profile of a whole step when there is one (inside the timed
    # region only this kernel carries events)
    fg = kernel_groups(full_prof, precision) if full_prof else groups
    share = (fg[key][2] if key in fg else ms) / sum(g[2] for g in fg.values())
    peak_tf = PEAK_TFLOPS[precision]
    t_mfma, t_hbm = flops / (peak_tf * 1e12), nbytes / (HBM_PEAK_GBS * 1e9)
    kname = key[0] + key[1]
    if t_mfma >= t_hbm:
        ach = flops / (ms * 1e-3) / 1e12
        out = {"bound": "mfma", "achieved": ach, "peak": peak_tf, "unit": "TFLOP/s", "frac": ach / peak_tf}
    else:
        ach = nbytes / (ms * 1e-3) / 1e9
        out = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
    out.update({"traffic": pmc_traffic(kname.split(" ")[0].replace(",", ", ")), "kernel": kname, "launches": n,
                "avg_launch_ms": ms / n, "avg_gflop_per_launch": flops / n / 1e9,
                "avg_algorithmic_mb_per_launch": nbytes / n / 1e6,
                "time_share_of_conv": share})
    return out


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
    return {"value": batch / t_train, "unit": "patches/s", "cores": torch.get_num_threads(), "cpu_model": cpu_model(),
            "kind": "port",
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
                    help="arithmetic of the 3x3x3 convolutions (default: exact fp32, the BASELINE cfg2 mode; bf16 / fp16: "
                         "16-bit operands, fp32 accumulate -- BASELINE cfg3 / cfg5)")
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
    ddp_kw = {"tail_bucket_bytes": int(os.environ["M355_DDP_TAIL"])} if "M355_DDP_TAIL" in os.environ else {}   # (A/B hook)
    runner = D.PatchParallel(model, force_collectives=force_ddp, **ddp_kw) if (world > 1 or force_ddp) else model
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

    # The last warm-up step is event-timed launch by launch (per-op summary + which kernel dominates); inside the
    # timed region only the dominant kernel's launches carry timing events (the roofline measurement), so the
    # timed steps are not perturbed by ~110 events each.
    prof_w = None
    for i in range(args.warmup):
        if i == args.warmup - 1 and rank == 0:
            ops.CONV_PROFILE = []
        train_step(runner, crit, opt, predictor, batch, device)
    if ops.CONV_PROFILE is not None:
        torch.cuda.synchronize()
        prof_w, ops.CONV_PROFILE = ops.CONV_PROFILE, None
        ops.CONV_PROFILE_KEYS, ops.CONV_PROFILE_EVERY = dominant_keys(prof_w, args.precision)
    # A full (generation-2) collection of CPython's cyclic GC walks every object alive -- ~90 ms with torch
    # imported -- and lands at an arbitrary step (measured with tools/step_trace.py: one such host stall drains
    # the launch queue and idles the GPU for ~8 ms; on a 12 ms step that is +2..4 ms/step of noise in a 10-20 step
    # window).  Everything allocated so far is long-lived: move it out of the collector's reach, as any
    # long-running service does; the young generations still collect the per-step garbage.
    gc.collect()
    gc.freeze()
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
    infer, prof_inf = None, None
    if not args.no_infer:
        model.eval()
        with torch.no_grad():
            ops.CONV_PROFILE_KEYS, ops.CONV_PROFILE_EVERY = None, 1
            for i in range(max(1, args.warmup)):
                if i == max(1, args.warmup) - 1 and rank == 0:
                    ops.CONV_PROFILE = []
                model(x)
            if ops.CONV_PROFILE is not None:
                torch.cuda.synchronize()
                ops.CONV_PROFILE_KEYS, _ = dominant_keys(ops.CONV_PROFILE, args.precision)
            ops.CONV_PROFILE = None
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                model(x)
            barrier()
            ti = time.perf_counter() - t0
            # roofline of the inference path: the same K forwards once more with the dominant kernel's launches
            # event-timed (outside the timed loop: in the 16-bit modes ~20 launches of a 2.3 ms forward carry events)
            ops.CONV_PROFILE = [] if rank == 0 else None
            for _ in range(args.steps):
                model(x)
            torch.cuda.synchronize()
            prof_inf, ops.CONV_PROFILE = ops.CONV_PROFILE, None
            ops.CONV_PROFILE_KEYS = None
        ti_t = torch.tensor([ti], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(ti_t, op=dist.ReduceOp.MAX)
        ti = float(ti_t.item())
        infer = {"value": world * args.batch * args.steps / ti, "unit": "patches/s", "ms_per_step": ti / args.steps * 1e3}

    if rank == 0:
        # ---- roofline of the dominant conv kernel of the timed train region (and of the inference region) ----
        roofline = roofline_of(prof, args.precision, prof_w)
        if infer is not None and prof_inf:
            infer["roofline"] = roofline_of(prof_inf, args.precision)
        # per-op summary: of the fully profiled warm-up step when there was one, else of the timed region
        conv_summary = conv_summary_of(prof_w, 1) if prof_w else conv_summary_of(prof, args.steps)

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
