#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r04
timeout -k 10 300 python tools/arch_bench.py all fp32 > gpurun_out/r04/arch_fp32.log 2>&1; echo rc=$?; grep -v amdgpu.ids gpurun_out/r04/arch_fp32.log
timeout -k 10 300 python tools/arch_bench.py all fp32_mfma > gpurun_out/r04/arch_fp32_mfma.log 2>&1; echo rc=$?; grep -v amdgpu.ids gpurun_out/r04/arch_fp32_mfma.log
