#!/usr/bin/env python3
"""Copy the summaries of tools/r02_measure.sh (gpurun_out/r02, gpurun_out/pmc) into profiles/r02_*."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "r02"), os.path.join(ROOT, "profiles")


def last_json(log, out):
    path = os.path.join(SRC, log)
    if not os.path.exists(path):
        return
    line = [l for l in open(path) if l.startswith("{")][-1]
    with open(os.path.join(DST, out), "w") as f:
        json.dump(json.loads(line), f, indent=1)
        f.write("\n")


def text(log, out):
    path = os.path.join(SRC, log)
    if not os.path.exists(path):
        return
    lines = [l for l in open(path) if "amdgpu.ids" not in l]
    with open(os.path.join(DST, out), "w") as f:
        f.writelines(lines)


which = sys.argv[1:] or ["a", "b"]
if "a" in which:
    last_json("bench_fp32.log", "r02_final_bench.json")
    last_json("bench_bf16.log", "r02_bf16_bench.json")
    last_json("bench_fp16.log", "r02_fp16_bench.json")
    for name, out in (("prof_fp32", "r02_final_kernel_stats.csv"), ("prof_bf16", "r02_bf16_kernel_stats.csv"),
                      ("prof_bf16_infer", "r02_bf16_infer_kernel_stats.csv")):
        p = os.path.join(SRC, name + "_kernel_stats.csv")
        if os.path.exists(p):
            shutil.copy(p, os.path.join(DST, out))
if "b" in which:
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summarize.py"), os.path.join(ROOT, "gpurun_out", "pmc"),
                           os.path.join(DST, "r02_pmc_traffic.json")], stdout=subprocess.DEVNULL)
    text("arch.log", "r02_arch_bench.txt")
    text("arch_bf16.log", "r02_arch_bench_bf16.txt")
    text("layers_cfg2.log", "r02_cfg2_fp32_per_layer.txt")
    text("layers_msseg2.log", "r02_msseg2_per_layer.txt")
    text("conv_fp32.log", "r02_fp32_conv_per_layer.txt")
    text("conv_bf16.log", "r02_bf16_c8_conv_per_layer.txt")
    text("sliding.log", "r02_cfg4_sliding_window_phases.txt")
    text("bww_classes.log", "r02_bww_pair_classes.txt")
    text("layers_dmri.log", "r02_dmri_hippo_per_layer.txt")
    text("train_breakdown.log", "r02_train_step_breakdown.txt")
    text("arch_breakdown.log", "r02_arch_kernel_breakdown.txt")
print("profiles/:", " ".join(sorted(f for f in os.listdir(DST) if f.startswith("r02_"))))
