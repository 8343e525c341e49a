#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r04
bash tools/train_breakdown.sh fp32 > gpurun_out/r04/breakdown_fp32.txt 2>&1; echo rc=$?
cat gpurun_out/r04/breakdown_fp32.txt
python3 tools/r04/topk.py "$(find gpurun_out/breakdown_fp32 -name s_kernel_stats.csv | head -1)" 12 45 > gpurun_out/r04/breakdown_fp32_top.txt 2>&1; cat gpurun_out/r04/breakdown_fp32_top.txt
cp "$(find gpurun_out/breakdown_fp32 -name 's_kernel_stats.csv' | head -1)" gpurun_out/r04/breakdown_fp32_kernel_stats.csv
