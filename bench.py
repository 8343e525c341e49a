#!/usr/bin/env python3
"""Benchmark of the hot path: BASELINE.json's metric on its cfg2 workload.

    python bench.py --gpus N --steps K --warmup W          (any N: with N > 1 and no WORLD_SIZE in the
                                                             environment it starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N --workload cfg3               (cfg2 network, bf16 operands: BASELINE cfg3)
    python bench.py --gpus N --workload cfg4               (sliding-window inference, tiles sharded over the ranks)

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
PEAK_TFLOPS = {"fp32": 157.3, "fp32_mfma": 157.3, "bf16": 2500.0, "fp16": 2500.0}
# precision "fp32": the split kernels execute six bf16 MFMAs per fp32 product group (exact 3-way operand split) and 28
# tap slots for the 27 taps (forward / data gradient: 14 tap pairs; weight gradient: 7 taps on each of 4 waves)
X3_KERNELS, X3_MFMA_FLOPS_PER_FLOP = ("conv3_f32x3_kernel", "conv3_bww_x3_kernel"), 6.0 * 28.0 / 27.0
HBM_PEAK_GBS = 8000.0
# m355_conv3d_plan(): kernel family -> kernel name
PLAN_KERNEL = {1: "conv3_mfma_fwd_kernel", 3: "conv3_mfma_fwd_p_kernel", 2: "conv3_valu_smallcout_kernel",
               4: "conv3_h16_kernel", 5: "conv3_h16_kernel(8 waves)", 6: "conv3_h16_kernel(one-shot)", 7: "conv3_f32x3_kernel",
               8: "conv3_bww_x3_kernel", 9: "conv3_mfma_bww2_kernel", 10: "conv3_mfma_bww_small_kernel", 11: "conv3_bww_c8_kernel",
               0: "conv3d_direct_kernel"}
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


def pmc_traffic(kernel_substr, precision="fp32"):
    """HBM bytes per launch of a kernel from the PMC passes of this same command (tools/pmc_collect.sh ->
    profiles/r04_pmc_traffic[_bf16|_fp16].json: separate FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 fetch correction).  PMC
    counters cannot be read inside the run; the committed figure is only quoted when it was collected on
    exactly these kernel sources (source_hash), otherwise null."""
    path = os.path.join(ROOT, "profiles", "r04_pmc_traffic.json" if precision == "fp32" else f"r04_pmc_traffic_{precision}.json")
    try:
        with open(path) as f:
            doc = json.load(f)
        if doc.get("source_hash") != source_hash():
            return None
        # "conv3_mfma_fwd_p_kernel<4, 32>" also has to find "...<4, 32, false>" (trailing template arguments); the
        # 16-bit kernels are reported with their type spelled out ("conv3_h16_kernel<4, 32, __bf16, true, 4, true>"):
        # "conv3_h16_kernel(one-shot)<4,32>" -> name stem + leading template arguments
        base = kernel_substr.split("(")[0].split("<")[0]
        targs = kernel_substr[kernel_substr.index("<") + 1:].rstrip(">") if "<" in kernel_substr else ""
        best = None
        for name, rec in doc["kernels"].items():
            if base not in name:
                continue
            if targs and not (f"<{targs}>" in name or f"<{targs}," in name):
                continue
            if best is None or rec["launches"] > best["launches"]:
                best = rec
        return None if best is None else best["hbm_bytes_per_launch"]
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
        if tag == "conv3d_bwd_weight" and plan is not None and plan[0]:   # fp32-tensor entry point (m355_conv3d_plan which = 2)
            key = (PLAN_KERNEL.get(plan[0], "conv3d_bwd_weight"), f"<{plan[2]}>")
        elif tag == "conv3d_bwd_weight":                                   # the c8 entry points of the 16-bit training flow
            key = ("conv3_bww_c8_kernel", "")
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


def roofline_of(prof, precision, full_prof=None, traffic=True):
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
    alg_flops = flops
    if key[0] in X3_KERNELS:   # what the matrix core executes, against the peak of the instruction it executes
        flops, peak_tf = flops * X3_MFMA_FLOPS_PER_FLOP, PEAK_TFLOPS["bf16"]
    t_mfma, t_hbm = flops / (peak_tf * 1e12), nbytes / (HBM_PEAK_GBS * 1e9)
    kname = key[0] + key[1]
    if t_mfma >= t_hbm:
        ach = flops / (ms * 1e-3) / 1e12
        out = {"bound": "mfma", "achieved": ach, "peak": peak_tf, "unit": "TFLOP/s", "frac": ach / peak_tf}
        if key[0] in X3_KERNELS:
            out.update({"algorithmic_tflops": alg_flops / (ms * 1e-3) / 1e12, "mfma_flops_per_algorithmic_flop": X3_MFMA_FLOPS_PER_FLOP,
                        "fp32_mfma_peak_tflops": PEAK_TFLOPS["fp32_mfma"],
                        "note": "achieved / peak: bf16 MFMA flops EXECUTED (six plane products per fp32 product, 28 tap slots "
                                "for 27 taps) against the dense bf16 peak; algorithmic_tflops: 2*27*Cin*Cout*voxels / time"})
    else:
        ach = nbytes / (ms * 1e-3) / 1e9
        out = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
    # (the committed PMC passes are of the cfg2 command: quoted only for launches of that workload)
    out.update({"traffic": pmc_traffic(kname.split(" ")[0].replace(",", ", "), precision) if traffic else None, "kernel": kname, "launches": n,
                "avg_launch_ms": ms / n, "avg_gflop_per_launch": flops / n / 1e9,
                "avg_algorithmic_mb_per_launch": nbytes / n / 1e6,
                "time_share_of_conv": share})
    return out


def _timed(fn, warmup=1, reps=3):
    """BASELINE.md section 3 / SURVEY section 8d protocol for the CPU leg: `warmup` untimed + `reps` timed iterations -> (mean s, last result)"""
    out = None
    for _ in range(warmup):
        out = fn()
    t0 = time.time()
    for _ in range(reps):
        out = fn()
    return (time.time() - t0) / reps, out


def cpu_baseline(cfg, batch, train=True, n_out=None):
    """Stock torch-CPU restatement of the reference path (oracle/torch_ref.py, pinned to the real
    reference by tests/golden) on a bounded sample of the same workload on the host cores: 1 warm-up + 3 timed
    no-grad forwards and (train=True) 1 warm-up + 3 timed train steps."""
    from oracle import torch_ref as R
    cin, cout, filters, depth, patch = cfg
    cout = cout if n_out is None else n_out
    torch.manual_seed(0)
    model = build_model((cin, cout, filters, depth, patch))
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd0.items()}
    spec = R.UNetSpec(cin, cout, filters, depth, norm="group", groups=8, up="convT")
    g = torch.Generator().manual_seed(1234)
    x = torch.randn((batch, cin) + patch, generator=g)
    lab = torch.randint(0, cout, (batch,) + patch, generator=g)
    y = torch.nn.functional.one_hot(lab, cout).permute(0, 4, 1, 2, 3).float().contiguous()
    # the GPU box gives one GPU a share of 16 host cores (os.cpu_count() reports the whole host)
    cores = min(os.cpu_count() or 1, int(os.environ.get("M355_CPU_THREADS", "16")))
    torch.set_num_threads(cores)

    def infer():
        with torch.no_grad():
            return R.unet_forward(sd, spec, x, training=False)
    t_inf, p0 = _timed(infer)
    # Dice of the seed-0 weights BEFORE any update (what the GPU's step-0 figures are compared with)
    with torch.no_grad():
        ld0 = R.hybrid_logistic_dice_loss(p0, y)
    out = {"unit": "patches/s", "cores": torch.get_num_threads(), "cpu_model": cpu_model(), "kind": "port",
           "infer_value": batch / t_inf, "dice_loss": float(ld0["dice_loss"]), "loss": float(ld0["loss"]), "_probs": p0.detach()}
    shape = f"{batch}x{cin}x{'x'.join(map(str, patch))}"
    if train:
        params = [v for v in sd.values() if v.requires_grad]
        opt = torch.optim.SGD(params, lr=1e-3, momentum=0.95)

        def step():
            p = R.unet_forward(sd, spec, x, training=True)
            ld = R.hybrid_logistic_dice_loss(p, y)
            opt.zero_grad()
            ld["loss"].backward()
            opt.step()
            return ld
        t_train, _ = _timed(step)
        out.update({"value": batch / t_train,
                    "sample": f"1 warm-up + 3 timed no-grad forwards ({t_inf:.2f} s each) and 1 warm-up + 3 timed train steps "
                              f"({t_train:.2f} s each) of the same {shape} workload, torch-CPU restatement of the reference"})
    else:
        out.update({"value": batch / t_inf,
                    "sample": f"1 warm-up + 3 timed no-grad forwards ({t_inf:.2f} s each) of one {shape} tile of the same "
                              f"sliding window, torch-CPU restatement of the reference"})
    return out


# ------------------------------------------------------------------------------------------ launcher
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes (one per GPU, torchrun's
    own elastic agent) before this process has made any GPU call, forward their output (rank 0 prints the JSON line)
    and return their exit code.  Nothing is re-exec'ed: this parent stays a plain CPU process."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC (RCCL across processes on this host driver)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=list(WORKLOADS) + ["cfg3", "cfg4"],
                    help="cfg2: the headline (fp32 train step + inference forward); cfg3 = cfg2 with --precision bf16; "
                         "cfg4: sliding-window inference of a 4x256^3 volume (patch 160, overlap 20), tiles sharded over the ranks")
    ap.add_argument("--batch", type=int, default=1, help="patches per rank per step")
    ap.add_argument("--precision", default=None, choices=["fp32", "fp32_mfma", "bf16", "fp16"],
                    help="arithmetic of the 3x3x3 convolutions (default fp32, the BASELINE cfg2 mode: fp32 tensors and fp32 "
                         "accuracy, the products of the wide layers as six bf16 MFMAs on an exact 3-way operand split; "
                         "fp32_mfma: every product on the fp32 MFMA; bf16 / fp16: 16-bit operands, fp32 accumulate -- "
                         "BASELINE cfg3 / cfg5)")
    ap.add_argument("--bucket-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="wire type of the gradient all-reduce buckets (bf16 halves the RCCL volume; fp32 master gradients)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-infer", action="store_true")
    ap.add_argument("--no-cfg3", action="store_true", help="skip the extra bf16 (BASELINE cfg3) and fp32_mfma measurements of the cfg2 / fp32 line")
    ap.add_argument("--launch-probe", default=None, help=argparse.SUPPRESS)   # tests: each rank writes its env here and exits
    args = ap.parse_args(argv)
    if args.workload == "cfg3":
        args.workload, args.precision = "cfg2", args.precision or "bf16"
    args.precision = args.precision or "fp32"
    return args


def rccl_info(world):
    if not dist.is_initialized():
        return {"ranks": 1, "backend": None, "version": None}
    try:
        ver = ".".join(map(str, torch.cuda.nccl.version()))
    except Exception:   # noqa: BLE001 -- informational only
        ver = None
    return {"ranks": dist.get_world_size(), "backend": dist.get_backend(), "version": ver}


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: become one (before any GPU call in this process)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_probe:
        with open(os.path.join(args.launch_probe, f"rank{rank}.json"), "w") as f:
            json.dump({"rank": rank, "world": world, "local_rank": local_rank, "gpus": args.gpus,
                       "master": os.environ.get("MASTER_ADDR")}, f)
        return
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks; measuring {world}", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_ddp = os.environ.get("M355_FORCE_DDP", "0") == "1"  # exercise the RCCL path with one rank
    if world > 1 or force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import segmentation_pipeline_amd as sp
    sp.set_precision(args.precision)
    if args.workload == "cfg4":
        out = run_window(args, rank, world, device)
    else:
        out = run_train(args, rank, world, device, force_ddp)
        if args.workload == "cfg2" and args.precision == "fp32" and not args.no_cfg3:
            # BASELINE cfg3 (same net, bf16 operands) under the same clock as the headline: K train steps + K forwards
            torch.cuda.empty_cache()
            c3 = run_train(args, rank, world, device, force_ddp, precision="bf16")
            sp.set_precision(args.precision)
            # the same fp32 workload with every convolution product on the fp32 MFMA (the arithmetic of rounds 1-3)
            torch.cuda.empty_cache()
            mf = run_train(args, rank, world, device, force_ddp, precision="fp32_mfma")
            sp.set_precision(args.precision)
            if rank == 0:
                out["cfg3"] = c3
                out["fp32_mfma"] = mf
                mf["max_abs_diff_of_step0_probabilities_vs_headline"] = float((mf.pop("_p0") - out.pop("_p0")).abs().max())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    # The CPU leg runs on rank 0 after the process group is gone (the other ranks have left; no collective can time
    # out behind ~30 s of host work) and for every N, so each line of a scaling run carries its own baseline.
    if not args.no_cpu_baseline:
        finish = out.pop("_cpu_leg")
        finish(out)
    out.pop("_cpu_leg", None)
    print(json.dumps(out), flush=True)


def barrier():
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(seconds, device):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def dtype_label(precision):
    from segmentation_pipeline_amd import ops
    if precision == "fp32" and ops.FP32_SPLIT:
        return ("f32 (tensors, results and accumulation; the products of the 3x3x3 convolutions with >= 8 input channels run as "
                "six bf16 MFMAs on an EXACT three-way bf16 split of both f32 operands -- error vs f64 equal to the f32 MFMA "
                "kernels', profiles/r04_f32x3_accuracy.txt; the same step on v_mfma_f32_32x32x2_f32 is the \"fp32_mfma\" key)")
    if precision in ("fp32", "fp32_mfma"):
        return "f32"
    return f"{precision} operands / f32 accumulate (3x3x3 convs), f32 elsewhere"


def run_train(args, rank, world, device, force_ddp, precision=None):
    """precision=None: the workload's own line (args.precision).  precision="bf16" on a cfg2 / fp32 invocation: the
    BASELINE cfg3 numbers (same network, same patch, 16-bit operands) measured by the SAME command after the fp32
    timed region, returned as the extra "cfg3" object of the same JSON line (so the driver's clock covers them)."""
    from segmentation_pipeline_amd import distributed as D
    from segmentation_pipeline_amd import ops
    from segmentation_pipeline_amd.criterions import HybridLogisticDiceLoss
    from segmentation_pipeline_amd.prediction import StandardPredict
    from segmentation_pipeline_amd.trainer import PhaseTimer, hard_dice_from_counts, train_step

    import segmentation_pipeline_amd as sp
    extra = precision is not None
    precision = precision or args.precision
    sp.set_precision(precision)
    cfg = WORKLOADS[args.workload]
    cin, cout, filters, depth, patch = cfg
    model = build_model(cfg).to(device)
    crit = HybridLogisticDiceLoss()
    # research/msseg2/msseg2.py:94.  fp16: the FUSED flavour of the same optimizer, whose kernels take the overflow word of
    # the loss scaling as `found_inf` on the device (trainer.train_step) -- the foreach flavour needs a host read per step
    opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.95, **({"fused": True} if precision == "fp16" else {}))
    ddp_kw = {"tail_bucket_bytes": int(os.environ["M355_DDP_TAIL"])} if "M355_DDP_TAIL" in os.environ else {}   # (A/B hook)
    if args.bucket_dtype == "bf16":
        ddp_kw["bucket_dtype"] = torch.bfloat16
    runner = D.PatchParallel(model, force_collectives=force_ddp, **ddp_kw) if (world > 1 or force_ddp) else model
    predictor = StandardPredict(image_names=["X", "y"])
    x, lab, y = synth((args.batch, cin) + patch, cout, 1234 + rank, device)
    batch = {"X": x, "y": y}

    # ---- Dice vs CPU reference: first forward from the seed-0 weights, before any update ----
    model.eval()
    with torch.no_grad():
        p0 = model(x)
    am, counts = ops.argmax_confusion(p0, lab.to(torch.int32))
    gpu_dice0 = float(crit(p0, y)["dice_loss"])
    hard0 = hard_dice_from_counts(counts)[0].tolist()
    p0_cpu, am_cpu_gpu, lab_cpu = (p0.cpu(), am.cpu().long(), lab.cpu()) if rank == 0 else (None, None, None)

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
        ops.CONV_PROFILE_KEYS, ops.CONV_PROFILE_EVERY = dominant_keys(prof_w, precision)
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
    elapsed = max_over_ranks(elapsed, device)
    value = world * args.batch * args.steps / elapsed

    # per-phase breakdown (TorchTimer semantics: a sync per stamp), outside the timed region
    phases = None
    if not extra:
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
                ops.CONV_PROFILE_KEYS, _ = dominant_keys(ops.CONV_PROFILE, precision)
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
        ti = max_over_ranks(ti, device)
        infer = {"value": world * args.batch * args.steps / ti, "unit": "patches/s", "ms_per_step": ti / args.steps * 1e3}

    gc.unfreeze()
    if rank != 0:
        return None
    # ---- roofline of the dominant conv kernel of the timed train region (and of the inference region) ----
    roofline = roofline_of(prof, precision, prof_w)
    if infer is not None and prof_inf:
        infer["roofline"] = roofline_of(prof_inf, precision)
    if extra:
        # the cfg3 object of the headline line: same keys as a line of its own, without the CPU leg / Dice block
        return {"ms_per_step": elapsed / args.steps * 1e3, "value": value, "unit": "patches/s",
                "infer_ms": infer["ms_per_step"] if infer else None, "infer_value": infer["value"] if infer else None,
                "steps": args.steps, "warmup": args.warmup, "dtype": dtype_label(precision), "roofline": roofline,
                "infer_roofline": infer.get("roofline") if infer else None,
                "conv_kernels": conv_summary_of(prof_w, 1) if prof_w else None,
                "final_loss": float(loss_dict["loss"].detach()),
                "gpu_soft_dice_loss_step0": gpu_dice0, "gpu_hard_dice_step0": hard0,
                **({"_p0": p0} if precision == "fp32_mfma" else {}),
                "workload": ("the cfg2 workload itself with every convolution product on the fp32 MFMA (v_mfma_f32_32x32x2_f32), "
                             "timed by this same command after the headline region" if precision == "fp32_mfma" else
                             "BASELINE cfg3 on this rank count: the cfg2 network and patch with bf16 conv operands "
                             "(fp32 accumulate), timed by this same command after the fp32 region")}
    # per-op summary: of the fully profiled warm-up step when there was one, else of the timed region
    conv_summary = conv_summary_of(prof_w, 1) if prof_w else conv_summary_of(prof, args.steps)
    wire = 2 if args.bucket_dtype == "bf16" else 4
    n_params = sum(p.numel() for p in model.parameters())
    out = {
        "metric": METRIC, "value": value, "unit": "patches/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": dtype_label(precision), "data": "synthetic",
        "config": {"workload": f"{args.workload}: train step (fwd+loss+bwd+SGD) of ModularUNet({cin},{cout},"
                               f"{str(filters).replace(' ', '')},{depth},GroupNorm(8),ConvTranspose3d k2s2) on {args.batch}x{cin}x{'x'.join(map(str, patch))} per GPU",
                   "global_batch": world * args.batch, "params": n_params,
                   "parallelism": f"patch-parallel dp{world}" if world > 1 else "single GPU",
                   "optimizer": "torch.optim.SGD(lr=1e-3, momentum=0.95" + (", fused=True)" if precision == "fp16" else ")")},
        "infer": infer, "phases_ms": phases, "conv_kernels": conv_summary, "roofline": roofline,
        "rccl": dict(rccl_info(world), gradient_bytes_per_step=(n_params * wire if (world > 1 or force_ddp) else 0),
                     bucket_dtype=args.bucket_dtype),
        "final_loss": float(loss_dict["loss"].detach()),
        "dice": {"gpu_soft_dice_loss_step0": gpu_dice0, "gpu_hard_dice_step0": hard0},
    }
    if args.workload == "cfg2" and precision == "fp32" and not args.no_cfg3:
        out["_p0"] = p0   # compared with the fp32_mfma leg's first forward (main)

    def cpu_leg(out):
        cb = cpu_baseline(cfg, args.batch)
        p_cpu = cb.pop("_probs")
        # Dice vs CPU ref on identical synthetic volume and identical seed-0 weights
        am_cpu = p_cpu.argmax(dim=1)
        from oracle import torch_ref as R
        hard_cpu = [r[4] for r in R.hard_dice_table(am_cpu[0], lab_cpu[0], cout)]
        out["dice"].update({
            "cpu_soft_dice_loss_step0": cb["dice_loss"], "cpu_hard_dice_step0": hard_cpu,
            "max_abs_prob_diff_vs_cpu": float((p0_cpu - p_cpu).abs().max()),
            "argmax_mismatch_voxels": int((am_cpu_gpu != am_cpu).sum()),
        })
        out["cpu_baseline"] = cb
    out["_cpu_leg"] = cpu_leg
    return out


# cfg4: cfg2 network with 2 outputs, volume 4x256^3, patch 160, overlap 20, 'average'
CFG4 = {"volume": (4, 256, 256, 256), "patch": 160, "overlap": 20, "n_out": 2}


def run_window(args, rank, world, device):
    """One "step" = sliding-window inference of one resident 4x256^3 volume (research/msseg2/msseg2.py:139-146,
    prediction.py:124-152): 8 tiles of 4x160^3 through the model, averaged.  With N ranks the tiles of a volume are
    sharded (tile i -> rank i % N, one all_gather, every rank aggregates): total work is fixed -> "strong"."""
    from segmentation_pipeline_amd import distributed as D
    from segmentation_pipeline_amd import ops
    from segmentation_pipeline_amd.prediction import PatchPredict, grid_locations

    cin, _, filters, depth, _ = WORKLOADS["cfg2"]
    cfg = (cin, CFG4["n_out"], filters, depth, (CFG4["patch"],) * 3)
    model = build_model(cfg).to(device).eval()
    vol = torch.randn(CFG4["volume"], generator=torch.Generator().manual_seed(1234)).to(device)   # the SAME volume on every rank
    n_tiles = len(grid_locations(vol.shape[1:], (CFG4["patch"],) * 3, (CFG4["overlap"],) * 3))
    pp = PatchPredict(patch_batch_size=1, patch_size=CFG4["patch"], patch_overlap=CFG4["overlap"])
    with D.unit_sharding():
        for i in range(max(1, args.warmup)):
            if i == max(1, args.warmup) - 1 and rank == 0:
                ops.CONV_PROFILE = []
            out_vol = pp.predict_volume(model, vol)
        if ops.CONV_PROFILE is not None:
            torch.cuda.synchronize()
            ops.CONV_PROFILE_KEYS, ops.CONV_PROFILE_EVERY = dominant_keys(ops.CONV_PROFILE, args.precision)
            ops.CONV_PROFILE = None
        gc.collect()
        gc.freeze()
        ops.CONV_PROFILE = [] if rank == 0 else None
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out_vol = pp.predict_volume(model, vol)
        barrier()
        elapsed = time.perf_counter() - t0
        prof, ops.CONV_PROFILE = ops.CONV_PROFILE, None
        ops.CONV_PROFILE_KEYS = None
        elapsed = max_over_ranks(elapsed, device)
        # phase table (a device synchronisation per stamp), outside the timed region
        timings = {}
        pt = PatchPredict(patch_batch_size=1, patch_size=CFG4["patch"], patch_overlap=CFG4["overlap"], timings=timings)
        for _ in range(2):
            pt.predict_volume(model, vol)
    s = out_vol.sum(0)
    sum_err = float((s - 1).abs().max())
    digest = float(out_vol.double().sum())     # identical on every rank count (aggregation is in grid order)
    if rank != 0:
        return None
    phases = {k: v / 2 * 1e3 for k, v in timings.items()}
    serial = sum(v for k, v in phases.items() if k in ("exchange", "accumulate", "finalize", "aggregate"))
    out = {
        "metric": METRIC, "value": n_tiles * args.steps / elapsed, "unit": "patches/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": dtype_label(args.precision), "data": "synthetic",
        "config": {"workload": f"cfg4: sliding-window inference of one {'x'.join(map(str, CFG4['volume']))} volume per step, patch "
                               f"{CFG4['patch']}, overlap {CFG4['overlap']}, average; {n_tiles} tiles of {cin}x{CFG4['patch']}^3 through "
                               f"ModularUNet({cin},{CFG4['n_out']},{str(filters).replace(' ', '')},{depth},GroupNorm(8),ConvTranspose3d k2s2)",
                   "tiles_per_volume": n_tiles, "parallelism": f"tiles sharded over {world} ranks, one all_gather" if world > 1 else "single GPU"},
        "phases_ms": phases, "serial_ms_per_volume": serial,
        "roofline": roofline_of(prof, args.precision, traffic=False) if prof else None,
        "rccl": rccl_info(world),
        "checks": {"max_abs_sum_p_minus_1": sum_err, "sum_of_probabilities": digest},
    }

    def cpu_leg(out):
        cb = cpu_baseline(cfg, 1, train=False)
        cb.pop("_probs")
        out["cpu_baseline"] = cb
    out["_cpu_leg"] = cpu_leg
    return out


if __name__ == "__main__":
    main()
