#!/usr/bin/env python3
"""Copy the summaries of tools/r04_measure.sh (gpurun_out/r04m, gpurun_out/pmc*) into profiles/r04_*."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "r04m"), os.path.join(ROOT, "profiles")


def last_json(log, out):
    path = os.path.join(SRC, log)
    if not os.path.exists(path):
        return
    lines = [l for l in open(path) if l.startswith("{")]
    if not lines:
        return
    with open(os.path.join(DST, out), "w") as f:
        json.dump(json.loads(lines[-1]), f, indent=1)
        f.write("\n")


def text(log, out):
    path = os.path.join(SRC, log)
    if not os.path.exists(path):
        return
    lines = [l for l in open(path) if "amdgpu.ids" not in l and "UserWarning" not in l and "Consider using tensor.detach" not in l
             and "return p.detach()" not in l]
    with open(os.path.join(DST, out), "w") as f:
        f.writelines(lines)


which = sys.argv[1:] or ["a", "b", "c"]
if "a" in which:
    for log, out in (("bench_fp32.log", "r04_final_bench.json"), ("bench_bf16.log", "r04_bf16_bench.json"),
                     ("bench_fp16.log", "r04_fp16_bench.json"), ("bench_cfg4.log", "r04_cfg4_bench.json"),
                     ("bench_cfg4_bf16.log", "r04_cfg4_bf16_bench.json"), ("bench_cfg3_rccl1.log", "r04_cfg3_rccl_world1_bench.json")):
        last_json(log, out)
    for name, out in (("prof_fp32", "r04_final_kernel_stats.csv"), ("prof_fp32_mfma", "r04_fp32_mfma_kernel_stats.csv"),
                      ("prof_bf16", "r04_bf16_kernel_stats.csv"),
                      ("prof_bf16_infer", "r04_bf16_infer_kernel_stats.csv")):
        p = os.path.join(SRC, name + "_kernel_stats.csv")
        if os.path.exists(p):
            shutil.copy(p, os.path.join(DST, out))
if "b" in which:
    for prec, d, out in (("fp32", "pmc", "r04_pmc_traffic.json"), ("fp32_mfma", "pmc_fp32_mfma", "r04_pmc_traffic_fp32_mfma.json"),
                         ("bf16", "pmc_bf16", "r04_pmc_traffic_bf16.json"),
                         ("fp16", "pmc_fp16", "r04_pmc_traffic_fp16.json")):
        root = os.path.join(ROOT, "gpurun_out", d)
        if os.path.isdir(root):
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summarize.py"), root, os.path.join(DST, out), prec],
                                  stdout=subprocess.DEVNULL)
if "c" in which:
    for log, out in (("arch.log", "r04_arch_bench.txt"), ("arch_fp32_mfma.log", "r04_arch_bench_fp32_mfma.txt"),
                     ("arch_bf16.log", "r04_arch_bench_bf16.txt"),
                     ("layers_cfg2.log", "r04_cfg2_fp32_per_layer.txt"), ("conv_fp32.log", "r04_fp32_conv_per_layer.txt"),
                     ("conv_fp32_mfma.log", "r04_fp32_mfma_conv_per_layer.txt"), ("x3_accuracy.log", "r04_f32x3_accuracy.txt"),
                     ("conv_bf16.log", "r04_bf16_c8_conv_per_layer.txt"), ("convt_c8.log", "r04_convt_c8_per_level.txt"),
                     ("sliding.log", "r04_cfg4_sliding_window_phases.txt"), ("c8_probe.log", "r04_c8_training_flow_accuracy.txt"),
                     ("host.log", "r04_host_enqueue_vs_step.txt"), ("train_breakdown.log", "r04_train_step_breakdown.txt")):
        text(log, out)
print("profiles/:", " ".join(sorted(f for f in os.listdir(DST) if f.startswith("r04_"))))
