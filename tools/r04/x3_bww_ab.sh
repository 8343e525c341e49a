#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r04
timeout -k 10 300 python tools/r04/x3_bww_check.py 2>&1 | grep -v amdgpu | tail -3
timeout -k 10 200 python tools/conv_bench.py bwd_weight --f32x3 2>&1 | grep -v amdgpu | tee gpurun_out/r04/x3_bww_ab.log
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "f32x3 or 3x3x3" 2>&1 | tail -2
